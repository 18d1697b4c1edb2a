// achip_waveparse.h -- what the wavefront-per-block parsers of the two-pass decoders share (lz4_decompress_v7.hip lz4_parse_wave_kernel,
// snappy_decompress_v5.hip snappy_parse_wave_kernel): the stream's bytes passing through a staging area in LDS, and the sink that writes a
// block's records into chunks of the arena.
#pragma once
#include "achip_seqexec.h"

namespace achip {

namespace wp {
constexpr int LZ4_STAGE = 352;     // an LZ4 sequence that starts within 64 positions and has at most one extension byte per length ends within 337 bytes
constexpr int SNAPPY_STAGE = 144;  // a Snappy run with its length in the tag (<= 60 bytes) that starts within 64 positions, and the 1- or 2-byte-offset copy behind it, end within 128 bytes
constexpr int SLAB = 2048;         // the staging area holds the stream's bytes [B0, B0 + SLAB + STAGE); it moves on by a slab at a time
}
// The stream's bytes for the windows: a linear piece of the stream in LDS that moves on by 2 KiB when the windows have walked through 2 KiB -- the
// bytes of the NEXT slab are requested when a slab arrives and sit in registers until then (a window consumes ~40 bytes: fifty windows later).
// Staging every window's STAGE bytes on its own cost a memory round trip per window: 2 us for ~7 sequences.
template <int STAGE>
struct WaveStage {
    static constexpr int CAP = wp::SLAB + STAGE;
    uint8_t* lds;
    const uint8_t* in;
    int32_t inLimit;
    int32_t b0;        // (uniform) stream position of lds[0]; -1: nothing staged
    u32x4 pend[2];     // this lane's 2 x 16 bytes of [b0 + CAP, b0 + CAP + SLAB)
    int lane;
    __device__ __forceinline__ u32x4 fetch(int32_t pos) const
    {
        return pos + 16 <= inLimit ? ld16(in + pos) : u32x4{0, 0, 0, 0};  // (what lies beyond the block is never looked at: windows stay 8 bytes clear of the end)
    }
    __device__ __forceinline__ void request()
    {
        pend[0] = fetch(b0 + CAP + 16 * lane);
        pend[1] = fetch(b0 + CAP + 16 * (lane + 64));
    }
    __device__ __forceinline__ void restart(int32_t base)  // synchronous: the first window, and a window far beyond what is staged (behind a long literal run)
    {
        wave_sync();
        b0 = base;
        for (int32_t i = lane; i < CAP / 16; i += 64) {
            *(u32x4*)(lds + 16 * i) = fetch(b0 + 16 * i);
        }
        request();
        wave_sync();
    }
    __device__ __forceinline__ void advance()  // by one slab: the tail moves to the front, the requested slab lands behind it, the next one is requested
    {
        wave_sync();
        u32x4 tail = {0, 0, 0, 0};
        if (lane < STAGE / 16) {
            tail = *(const u32x4*)(lds + wp::SLAB + 16 * lane);
        }
        wave_sync();
        if (lane < STAGE / 16) {
            *(u32x4*)(lds + 16 * lane) = tail;
        }
        *(u32x4*)(lds + STAGE + 16 * lane) = pend[0];
        *(u32x4*)(lds + STAGE + 16 * (lane + 64)) = pend[1];
        b0 += wp::SLAB;
        request();
        wave_sync();
    }
    // the window at `base` is readable at lds + (base - b0)
    __device__ __forceinline__ const uint8_t* window(int32_t base)
    {
        if (b0 < 0 || base < b0 || base - b0 >= 2 * wp::SLAB) {  // (uniform)
            restart(base);
        }
        else if (base - b0 >= wp::SLAB) {
            advance();
        }
        return lds + (base - b0);
    }
};
struct WaveRecordSink {  // the block's records: chunks of the arena, claimed one at a time
    sx::ArenaHeader* hdr;
    uint64_t* arena;
    int32_t maxChunks;
    int32_t firstChunk, chunk, fill, count;  // (uniform)
    bool fallback;                           // (uniform) the arena is exhausted: the ring decoder takes the block
    // A batch of n (<= 64, uniform) records: begin(n) makes room (false: nothing may be stored -- the arena is exhausted), store() places one record at index
    // idx (< n) of the batch -- a lane may store several --, end(n) moves on.
    int32_t fresh;  // (uniform) the chunk a batch under way reaches into, or -1
    __device__ __forceinline__ bool begin(int32_t n, int lane)
    {
        fresh = -1;
        if (n <= 0 || fallback) {  // (uniform)
            return false;
        }
        if (n > sx::CHUNK_RECS - fill) {  // (uniform) the batch reaches into a new chunk
            int32_t c = 0;
            if (lane == 0) {
                c = atomicAdd(&hdr->nextChunk, 1);
            }
            c = sx::wave_bcast(c, 0);
            if (c >= maxChunks) {
                fallback = true;
                return false;
            }
            if (chunk >= 0) {
                if (lane == 0) {
                    arena[(int64_t)chunk * sx::CHUNK_SLOTS + sx::CHUNK_RECS] = (uint64_t)(uint32_t)c;  // link
                }
            }
            else {
                firstChunk = c;
            }
            fresh = c;
        }
        return true;
    }
    __device__ __forceinline__ void store(uint64_t rec, bool valid, int32_t idx)
    {
        if (valid) {
            const int32_t slot = fill + idx;
            if (slot < sx::CHUNK_RECS) {
                arena[(int64_t)chunk * sx::CHUNK_SLOTS + slot] = rec;
            }
            else {
                arena[(int64_t)fresh * sx::CHUNK_SLOTS + (slot - sx::CHUNK_RECS)] = rec;
            }
        }
    }
    __device__ __forceinline__ void end(int32_t n)
    {
        if (fresh >= 0) {
            chunk = fresh;
            fill = fill + n - sx::CHUNK_RECS;
        }
        else {
            fill += n;
        }
        count += n;
    }
    // n (<= 64, uniform) records, lane i's at index idx (< n) of the batch if `valid`
    __device__ __forceinline__ void put(uint64_t rec, bool valid, int32_t idx, int32_t n, int lane)
    {
        if (begin(n, lane)) {
            store(rec, valid, idx);
            end(n);
        }
    }
};

// The chain of a window: lane p holds `next` = where the sequence after the one at p would begin (>= 64: beyond the window), `stop` = p cannot be on the chain.
// members = the positions 0 -> next[0] -> next[next[0]] ... up to, not including, the first one that stops or lies beyond the window; cur = that position.
// Four hops per trip of the scalar loop: a hop is a lane read whose lane index was itself just read -- 111 clocks each, half of a window's time, when walked one
// by one (profiles/r05_notes.md section 12) --, so every lane first gathers where two, three and four hops lead (three lane gathers, all lanes at once), the loop
// reads the four of the current position side by side and marks them by compares.  (A stopping position ends the chain by pointing beyond the window; it is
// marked like any other and taken off afterwards.)
__device__ __forceinline__ void wave_chain(int32_t next, bool stop, unsigned long long stopMask, int lane, unsigned long long& members, int32_t& cur)
{
    const int32_t n1 = stop ? 64 : next;
    int32_t n2 = __shfl(n1, n1 & 63);
    n2 = n1 < 64 ? n2 : n1;
    int32_t n3 = __shfl(n1, n2 & 63);
    n3 = n2 < 64 ? n3 : n2;
    int32_t n4 = __shfl(n2, n2 & 63);
    n4 = n2 < 64 ? n4 : n2;
    members = 0;
    cur = 0;
    while (cur < 64) {  // (uniform)
        const int32_t a1 = __builtin_amdgcn_readlane(n1, cur), a2 = __builtin_amdgcn_readlane(n2, cur), a3 = __builtin_amdgcn_readlane(n3, cur);
        const int32_t a4 = __builtin_amdgcn_readlane(n4, cur);
        members |= __ballot(lane == cur || lane == a1 || lane == a2 || lane == a3);  // (a position beyond the window is nobody's lane)
        cur = a4;
    }
    const unsigned long long hit = members & stopMask;
    if (hit != 0) {  // (uniform) nothing behind a stopping position was marked: it pointed beyond the window
        const int first = __builtin_ctzll(hit);
        members &= (1ull << first) - 1ull;
        cur = first;
    }
}

}  // namespace achip
