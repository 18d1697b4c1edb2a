// achip_seqexec.h -- sequence records and the wavefront-per-block sequence EXECUTOR (lz4_decompress_v7.hip and friends).
//
// The LZ77 decoders for text-like data are split in two:
//   parse    a lane per block walks the token grammar (serial by nature) and writes one 8-byte RECORD per sequence
//            {literal length, match length, offset, header bytes skipped} into a chunked arena -- no byte is copied;
//   execute  a WAVEFRONT per block runs 64 records at a time: two wave scans give every sequence its source and destination, the 64
//            literal runs are copied side by side, then the 64 matches -- those whose source lies in this batch's own output wait for
//            exactly the lanes that produce it (dependency masks), everything else goes at once.
// With a wavefront per block only a few thousand blocks are open at a time, so their 64 KiB windows stay in the L2 / Infinity Cache
// (the lane-per-block decoders keep 131072+ blocks open: 8x more memory traffic than bytes decoded on text), input and
// record streams are read coalesced, and the output is written in stream order.
//
// Cross-lane operations are only used in wave-uniform control flow and lanes exchange data through memory only across wave_sync():
// the kernels built on this header run unchanged under tools/hostemu (fibers) on a CPU.
#pragma once
#include "achip_device.h"

namespace achip {
namespace sx {

// ---- records --------------------------------------------------------------------------------------------------------------------
// bits  0..16  literal length   (<= 131071; longer runs are split over several records)
// bits 17..33  match length     (0 = none: the last literals of a block, or a split)
// bits 34..49  offset           (1..65535)
// bits 50..63  skip             compressed bytes between the end of the previous record's literals and this record's literals
//                               (tokens, length extensions, offsets; <= 16383, longer gaps are split)
constexpr int MAX_LEN = (1 << 17) - 1;
constexpr int MAX_SKIP = (1 << 14) - 1;
__device__ __forceinline__ uint64_t rec_pack(uint32_t lit, uint32_t ml, uint32_t off, uint32_t skip)
{
    return (uint64_t)lit | ((uint64_t)ml << 17) | ((uint64_t)off << 34) | ((uint64_t)skip << 50);
}
__device__ __forceinline__ int32_t rec_lit(uint64_t r) { return (int32_t)(r & 0x1FFFF); }
__device__ __forceinline__ int32_t rec_ml(uint64_t r) { return (int32_t)((r >> 17) & 0x1FFFF); }
__device__ __forceinline__ int32_t rec_off(uint64_t r) { return (int32_t)((r >> 34) & 0xFFFF); }
__device__ __forceinline__ int32_t rec_skip(uint64_t r) { return (int32_t)(r >> 50); }

// ---- arena: chunks of 512 slots (4 KiB); slots 0..510 hold records, slot 511 the index of the block's next chunk -------------------
constexpr int CHUNK_SLOTS = 512;
constexpr int CHUNK_RECS = CHUNK_SLOTS - 1;

struct BlockMeta {  // per block, written by the parser
    int32_t firstChunk;
    int32_t count;  // records to execute (0: nothing -- failed, empty or handed to the fallback decoder)
};

struct ArenaHeader {  // leads the scratch
    int32_t nextChunk;  // allocation cursor
    int32_t maxChunks;
    int32_t fallbackBlocks;  // blocks whose records did not fit: decoded by the ring decoder afterwards
    int32_t pad[61];
};

// ---- wave helpers (DPP on the device, shuffles under tools/hostemu) ---------------------------------------------------------------
__device__ __forceinline__ int32_t wave_scan_incl(int32_t x, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x += t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    x += t;
    const int32_t r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
    return x + (lane >= 16 ? r0 : 0) + (lane >= 32 ? r1 : 0) + (lane >= 48 ? r2 : 0);
#else
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t t = __shfl_up(x, d);
        if (lane >= d) x += t;
    }
    return x;
#endif
}
__device__ __forceinline__ int32_t wave_bcast(int32_t v, int srcLane) { return __builtin_amdgcn_readlane(v, srcLane); }
__device__ __forceinline__ const uint8_t* wave_bcast_ptr(const uint8_t* p, int srcLane)
{
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)wave_bcast((int32_t)(uint32_t)v, srcLane), hi = (uint32_t)wave_bcast((int32_t)(v >> 32), srcLane);
    return (const uint8_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// ---- lane-private exact copies ------------------------------------------------------------------------------------------------------
// n bytes (n < 16) of v to dst, nothing else written
__device__ __forceinline__ void store_exact16(uint8_t* dst, u32x4 v, int32_t n)
{
    const uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
    if (n & 8) st8(dst, lo);
    const uint64_t x8 = (n & 8) ? hi : lo;
    if (n & 4) st4(dst + (n & 8), (uint32_t)x8);
    const uint32_t x4 = (n & 4) ? (uint32_t)(x8 >> 32) : (uint32_t)x8;
    if (n & 2) st2(dst + (n & 12), x4);
    const uint32_t x2 = (n & 2) ? x4 >> 16 : x4;
    if (n & 1) dst[n & 14] = (uint8_t)x2;
}
// 16 bytes at src, of which the first n (1..16) are wanted; never reads at or beyond srcEnd
__device__ __forceinline__ u32x4 load_upto16(const uint8_t* src, int32_t n, const uint8_t* srcEnd)
{
    if (src + 16 <= srcEnd) {
        return ld16(src);
    }
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int i = 0; i < n && i < 16; i++) {
        w[i >> 2] |= (uint32_t)src[i] << (8 * (i & 3));
    }
    return u32x4{w[0], w[1], w[2], w[3]};
}
// n bytes src -> dst; the ranges do not overlap, or the source lies at least 16 bytes before the destination (forward 16-byte steps are
// then safe: every step reads bytes that are final).  Exact.  [src, src + n) is readable; nothing at or beyond srcEnd is read.
__device__ __forceinline__ void copy_fwd(uint8_t* dst, const uint8_t* src, int32_t n, const uint8_t* srcEnd)
{
    if (n <= 0) {
        return;
    }
    if (n < 16) {
        store_exact16(dst, load_upto16(src, n, srcEnd), n);
        return;
    }
    int32_t k = 0;
#pragma unroll 1
    for (; k + 16 <= n; k += 16) {
        st16(dst + k, ld16(src + k));
    }
    if (k < n) {
        st16(dst + n - 16, ld16(src + n - 16));  // the tail, overlapping what was just written (same bytes)
    }
}
// a match: ml bytes at dst repeat what lies `off` bytes before them (off >= 1; the source may run into the destination)
__device__ __forceinline__ void copy_match(uint8_t* dst, int32_t off, int32_t ml, const uint8_t* outEnd)
{
    if (off >= 16 || off >= ml) {
        copy_fwd(dst, dst - off, ml, outEnd);
        return;
    }
    // short period: one period first, then -- what is written repeats the period -- twice as far back each step
    int32_t c = 0, d = off;
#pragma unroll 1
    while (c < ml) {
        const int32_t n = d < ml - c ? d : ml - c;
        copy_fwd(dst + c, dst + c - d, n, outEnd);
        c += n;
        d += d;
    }
}

constexpr int BIG = 2048;  // copies longer than this are moved by the whole wavefront, one after the other

// the whole wavefront copies n bytes src -> dst (uniform arguments); ranges disjoint, or src + 1024 <= dst
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, int32_t n, int lane)
{
    const int32_t full = n & ~15;
    for (int32_t base = 0; base < full; base += 1024) {
        const int32_t k = base + lane * 16;
        if (k < full) {
            st16(dst + k, ld16(src + k));
        }
        wave_sync();  // a match may read what the previous round wrote
    }
    if (lane < (n & 15)) {
        dst[full + lane] = src[full + lane];
    }
}

// ---- the executor -------------------------------------------------------------------------------------------------------------------
// Runs `count` records of one block, starting at slot 0 of chunk `chunk`.  in / inLen: the block's compressed bytes (literal source);
// out / outCap: its output.  The records were validated by the parser: every literal range lies inside the input, every match source
// inside the output produced so far, the total inside outCap.
__device__ __forceinline__ void exec_block(const uint8_t* __restrict__ in, int32_t inLen, uint8_t* out, int32_t outCap, const uint64_t* __restrict__ arena,
                                           int32_t chunk, int32_t count, int lane)
{
    const uint8_t* const inEnd = in + inLen;
    const uint8_t* const outEnd = out + outCap;
    int32_t srcPos = 0;  // compressed position behind the previous record's literals
    int32_t outPos = 0;
    int32_t slot = 0;
    while (count > 0) {  // (uniform)
        const uint64_t* const c = arena + (int64_t)chunk * CHUNK_SLOTS;
        int32_t nb = CHUNK_RECS - slot;
        nb = nb < 64 ? nb : 64;
        nb = nb < count ? nb : count;
        uint64_t r = 0;
        if (lane < nb) {
            r = c[slot + lane];
        }
        const int32_t lit = rec_lit(r), ml = rec_ml(r), off = rec_off(r), skip = rec_skip(r);
        const int32_t tot = lit + ml, adv = skip + lit;
        const int32_t oEnd = wave_scan_incl(tot, lane), sEnd = wave_scan_incl(adv, lane);
        const int32_t dstLit = outPos + oEnd - tot;
        const int32_t srcLit = srcPos + sEnd - lit;
        const int32_t dstM = dstLit + lit;
        const int32_t total = wave_bcast(oEnd, 63), sTotal = wave_bcast(sEnd, 63);

        // ---- literal runs: 64 side by side; the few long ones by the whole wavefront ----
        copy_fwd(out + dstLit, in + srcLit, lit > BIG ? 0 : lit, inEnd);
        for (unsigned long long m = __ballot(lit > BIG); m != 0; m &= m - 1) {  // (uniform)
            const int l = __builtin_ctzll(m);
            wave_copy(out + wave_bcast(dstLit, l), in + wave_bcast(srcLit, l), wave_bcast(lit, l), lane);
        }
        wave_sync();

        // ---- matches ----
        const int32_t span = ml < off ? ml : off;  // source bytes that are not this match's own output
        const int32_t srcM = dstM - off;
        bool pending = ml > 0;
        const bool big = ml > BIG && off >= 1024;  // long and far enough back: moved by the whole wavefront
        // everything whose source is older than this batch goes at once
        if (pending && !big && srcM + span <= outPos) {
            copy_match(out + dstM, off, ml, outEnd);
            pending = false;
        }
        const unsigned long long waiting = __ballot(pending);
        if (waiting != 0) {  // (uniform)
            wave_sync();
            // lane j waits for the still-pending lanes before it whose output [dstLit, dstM + ml) overlaps its source
            unsigned long long dep = 0;
            for (unsigned long long m = waiting; m != 0; m &= m - 1) {  // (uniform)
                const int p = __builtin_ctzll(m);
                const int32_t pS = wave_bcast(dstLit, p), pE = wave_bcast(dstM + ml, p);
                if (p < lane && pE > srcM && pS < srcM + span) {
                    dep |= 1ull << p;
                }
            }
            for (;;) {  // (uniform) the first pending lane is always ready
                const unsigned long long pm = __ballot(pending);
                if (pm == 0) {
                    break;
                }
                const bool ready = pending && (dep & pm) == 0;
                if (ready && !big) {
                    copy_match(out + dstM, off, ml, outEnd);
                }
                for (unsigned long long m = __ballot(ready && big); m != 0; m &= m - 1) {  // (uniform)
                    const int l = __builtin_ctzll(m);
                    const int32_t d = wave_bcast(dstM, l), o = wave_bcast(off, l), n = wave_bcast(ml, l);
                    wave_copy(out + d, out + d - o, n, lane);
                }
                pending = pending && !ready;
                wave_sync();
            }
        }
        wave_sync();
        outPos += total;
        srcPos += sTotal;
        slot += nb;
        count -= nb;
        if (slot == CHUNK_RECS && count > 0) {
            chunk = (int32_t)c[CHUNK_RECS];  // (uniform address: the link)
            slot = 0;
        }
    }
}

}  // namespace sx
}  // namespace achip
