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

// ---- arena: chunks of 512 slots (4 KiB); slots 0..503 hold records (an all-zero record does nothing), slot 504 the index of the
// block's next chunk ----
constexpr int CHUNK_SLOTS = 512;
constexpr int CHUNK_RECS = 504;  // (a multiple of 8: the parser writes records in 64-byte pieces; the link sits in slot 504)

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
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int srcLane)  // (uniform srcLane)
{
    return ((uint64_t)(uint32_t)wave_bcast((int32_t)(v >> 32), srcLane) << 32) | (uint32_t)wave_bcast((int32_t)(uint32_t)v, srcLane);
}
__device__ __forceinline__ const uint8_t* wave_bcast_ptr(const uint8_t* p, int srcLane)
{
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)wave_bcast((int32_t)(uint32_t)v, srcLane), hi = (uint32_t)wave_bcast((int32_t)(v >> 32), srcLane);
    return (const uint8_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// ---- the parser's record output: 8 records staged per lane in LDS, written as one 64-byte piece -----------------------------------------
// (one 8-byte store per trip and lane would be a partial write of its own cache line every time: measured 56 GB written for 8 GB of records)
struct RecordWriter {
    uint64_t* stage;  // this lane's LDS column: record k at stage[k * 64]
    int32_t firstChunk, chunk, fill, count, recFill;

    __device__ __forceinline__ void init(uint64_t* lds)
    {
        stage = lds;
        firstChunk = -1;
        chunk = -1;
        fill = CHUNK_RECS;
        count = 0;
        recFill = 0;
    }
    __device__ __forceinline__ void put(uint64_t r)
    {
        stage[recFill * 64] = r;
        recFill++;
    }
    // once per trip, in wave-uniform control flow: staged records leave when there are 8 of them (or the lane is done: padded with empty
    // records); a chunk for every lane that needs one -- one atomic per wavefront; an exhausted arena hands the block to the fallback
    template <int DBG>
    __device__ __forceinline__ void service(bool& done, bool& fallback, ArenaHeader* hdr, uint64_t* arena, int32_t maxChunks, int lane, int32_t flushAt = 8)
    {
        const bool flushDue = recFill >= flushAt || (done && recFill > 0);
        const bool need = flushDue && fill == CHUNK_RECS;
        const unsigned long long nm = __ballot(need);
        if (nm != 0) {  // (uniform)
            int32_t base = 0;
            if (lane == __builtin_ctzll(nm)) {
                base = atomicAdd(&hdr->nextChunk, (int32_t)__popcll(nm));
            }
            base = wave_bcast(base, __builtin_ctzll(nm));
            if (need) {
                const int32_t c = base + (int32_t)__popcll(nm & ((1ull << lane) - 1));
                if (c >= maxChunks) {
                    fallback = true;
                    done = true;
                    recFill = 0;
                }
                else {
                    if (chunk >= 0) {
                        arena[(int64_t)chunk * CHUNK_SLOTS + CHUNK_RECS] = (uint64_t)(uint32_t)c;  // link
                    }
                    else {
                        firstChunk = c;
                    }
                    chunk = c;
                    fill = 0;
                }
            }
        }
        if (flushDue && recFill > 0) {
            uint64_t r[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                r[k] = k < recFill ? stage[k * 64] : 0ull;
            }
            if (DBG != 1) {
                uint8_t* const dst = (uint8_t*)(arena + (int64_t)chunk * CHUNK_SLOTS + fill);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    st16(dst + 8 * k, u32x4{(uint32_t)r[k], (uint32_t)(r[k] >> 32), (uint32_t)r[k + 1], (uint32_t)(r[k + 1] >> 32)});
                }
            }
            fill += 8;
            count += 8;
            recFill = 0;
        }
    }
};

// ---- the parser's view of its compressed stream: a byte-addressable LDS ring per lane, fed by loads that are in flight for NS trips -----
// RING bytes of the stream are resident per lane, contiguous in LDS (the first 16 bytes once more behind the ring, so that any resident
// 16 bytes [v, v + 16) can be read at ring + (v % RING) without a wrap case), fed with aligned 32-byte granules.
// The point of the structure is WHEN the loads are waited for.  A wavefront's vector memory operations complete in order and are waited
// for by count, so a lane that refilled "when it ran dry, one piece ahead" made the whole wavefront wait for the load some other lane
// had issued a moment ago: one memory latency per trip (measured: 2.1 us per trip of ~200 instructions).  Here every trip has ONE place
// where granules are requested (`issue`, slot = trip mod NS) and ONE place where the granule requested NS trips earlier enters the ring
// (`land`, same slot): the loop is unrolled, the slots are distinct registers, and the wait the compiler places before `land` lets the
// NS - 1 younger requests stay in flight.  A lane whose bytes are not resident yet sits the trip out.
// "Virtual" positions count from the 32-byte aligned address at or before the stream's first byte.
template <int NS>
struct LaneFeed {
    static constexpr int RING = 128;
    static constexpr int GRAN = 32;
    static constexpr int STRIDE = RING + 16;  // (a multiple of 16: the granules are written as aligned 16-byte pieces)
    uint8_t* ring;
    const uint8_t* inAligned;
    const uint8_t* safe;  // what a lane that wants nothing loads from (the load itself is unconditional, see issue)
    int32_t inBase;    // virtual position of the stream's first byte (0..31)
    int32_t lastV;     // virtual position of the last 16-byte piece that holds a byte of the stream
    int32_t loadedV;   // (a multiple of 32) the ring holds virtual [.., loadedV)
    int32_t issueV;    // the next granule to request; loadedV + 32 * (granules in flight)
    u32x4 ga[NS], gb[NS];
    uint32_t validBits;  // bit s: slot s holds a granule of this lane (flags live in vector registers: no lane-mask bookkeeping at branches)

    // `anywhere`: an address (16-byte aligned, 16 bytes) that can always be read
    __device__ __forceinline__ void init(uint8_t* lds, int lane, const uint8_t* in, int32_t inLimit, const void* anywhere)
    {
        ring = lds + lane * STRIDE;
        inBase = (int32_t)((uintptr_t)in & (GRAN - 1));
        inAligned = in - inBase;
        lastV = inLimit > 0 ? ((inLimit + inBase - 1) & ~15) : 0;
        safe = (const uint8_t*)anywhere;
        loadedV = 0;
        issueV = 0;
        validBits = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            ga[s] = u32x4{0, 0, 0, 0};
            gb[s] = u32x4{0, 0, 0, 0};
        }
    }
    // the granule requested NS trips ago enters the ring
    template <int SLOT>
    __device__ __forceinline__ void land()
    {
        const uint32_t have = (validBits >> SLOT) & 1u;
        const int32_t slot = loadedV & (RING - 1);
        if (have != 0) {  // (only stores inside the branch: no state to merge behind it)
            *(u32x4*)(ring + slot) = ga[SLOT];
            *(u32x4*)(ring + slot + 16) = gb[SLOT];
            if (slot == 0) {
                *(u32x4*)(ring + RING) = ga[SLOT];
            }
        }
        loadedV += have != 0 ? GRAN : 0;
        validBits &= ~(1u << SLOT);
    }
    // Requests the next granule if the ring has room for it behind virtual position v (everything below v is consumed).  The load
    // instructions are executed by every lane on every trip -- lanes that want nothing read `safe`, all the same address -- so that the
    // compiler can count: with a request on every trip, "all but the NS - 1 youngest have completed" is what `land` has to wait for.  (A
    // request under a branch makes that count unknown and every wait a wait for everything.)  A 16-byte piece is read whole even where
    // the stream starts or ends inside it: the bytes around the stream lie in the same aligned 16 bytes, hence on the same page, and no
    // parse decision looks at them (positions at or beyond the stream's end only ever reach the general path, which reads the stream
    // itself).  Pieces entirely beyond the end repeat the last one.
    template <int SLOT>
    __device__ __forceinline__ void issue(int32_t v, bool active)
    {
        const bool want = active && issueV + GRAN - RING <= v;
        const int32_t a0 = issueV < lastV ? issueV : lastV;
        const int32_t a1 = issueV + 16 < lastV ? issueV + 16 : lastV;
        const uint8_t* p0 = want ? inAligned + a0 : safe;
        const uint8_t* p1 = want ? inAligned + a1 : safe;
        ga[SLOT] = *(const u32x4*)p0;
        gb[SLOT] = *(const u32x4*)p1;
        validBits |= want ? (1u << SLOT) : 0u;
        issueV += want ? GRAN : 0;
    }
    // the stream continues at virtual position v, beyond everything requested so far: what is in flight is dropped
    __device__ __forceinline__ void restart(int32_t v, bool doIt = true)
    {
        loadedV = doIt ? (v & ~(GRAN - 1)) : loadedV;
        issueV = doIt ? loadedV : issueV;
        validBits = doIt ? 0u : validBits;
    }
    __device__ __forceinline__ bool resident(int32_t v, int32_t n) const { return v + n <= loadedV; }
    __device__ __forceinline__ uint32_t rd32(int32_t v) const
    {
        uint32_t x;
        __builtin_memcpy(&x, ring + (v & (RING - 1)), 4);
        return x;
    }
};

// largest multiple of off (1..65535) that is <= x (off <= x <= 65535)
__device__ __forceinline__ int32_t largest_multiple(int32_t off, int32_t x)
{
    int32_t m = (int32_t)((float)x / (float)off);  // within one of the quotient (both below 2^24: exact operands)
    m -= m * off > x ? 1 : 0;
    m += (m + 1) * off <= x ? 1 : 0;
    return m * off;
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

// ---- the executor with an LDS output window -----------------------------------------------------------------------------------------
// The batch's output is composed in a per-wavefront LDS window first: byte-exact writes are cheap there, matches that read what the same
// batch (or the last batches) produced never leave the CU -- a dependency round costs an LDS round trip, not a trip to the L2 and back --
// and the window is drained to the output buffer in whole aligned 16-byte pieces, 1 KiB per store instruction.
//   win[p - winBase] holds output byte p for winBase <= p < outPos (a LINEAR buffer: no wrap-around cases in the copy paths; when a batch
//   would not fit behind outPos any more, the last SLIDE_KEEP bytes are moved to the front -- one 16-byte LDS read and write per lane);
//   bytes below flushPos are in the output buffer (flushPos >= winBase, so every byte is readable from the one or the other).
//   A batch takes as many records as fit CAP output bytes; a single record beyond that is moved straight between the global buffers by
//   the whole wavefront and the window restarts behind it.
constexpr int CAP = 1920;          // output bytes of one batch
// WIN: bytes of the window; SLIDE_KEEP: history kept when the window slides (what is older is read from the output buffer)

template <int WIN, int SLIDE_KEEP>
struct WinIo {
    uint8_t* win;  // LDS, WIN + 16 bytes
    uint8_t* out;
    int32_t winBase;   // (uniform) output position of win[0]
    int32_t flushPos;  // (uniform)

    // 16 bytes at output position p (only bytes below the write frontier are meaningful)
    __device__ __forceinline__ u32x4 read16(int32_t p) const
    {
        if (p >= winBase) {
            uint64_t a, b;
            __builtin_memcpy(&a, win + (p - winBase), 8);
            __builtin_memcpy(&b, win + (p - winBase) + 8, 8);
            return u32x4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
        }
        if (p + 16 <= winBase) {
            return ld16(out + p);  // flushed long ago
        }
        uint32_t w[4] = {0, 0, 0, 0};  // across winBase (cold)
#pragma unroll 1
        for (int k = 0; k < 16; k++) {
            const int32_t q = p + k;
            const uint32_t b = q >= winBase ? win[q - winBase] : out[q];
            w[k >> 2] |= b << (8 * (k & 3));
        }
        return u32x4{w[0], w[1], w[2], w[3]};
    }
    // the first n (0..16) bytes of v to output position p (inside the window)
    __device__ __forceinline__ void write_upto16(int32_t p, u32x4 v, int32_t n)
    {
        uint8_t* d = win + (p - winBase);
        const uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
        if (n >= 16) {
            __builtin_memcpy(d, &lo, 8);
            __builtin_memcpy(d + 8, &hi, 8);
            return;
        }
        if (n & 8) __builtin_memcpy(d, &lo, 8);
        const uint64_t x8 = (n & 8) ? hi : lo;
        const uint32_t x4lo = (uint32_t)x8;
        if (n & 4) __builtin_memcpy(d + (n & 8), &x4lo, 4);
        const uint32_t x4 = (n & 4) ? (uint32_t)(x8 >> 32) : (uint32_t)x8;
        const uint16_t x2lo = (uint16_t)x4;
        if (n & 2) __builtin_memcpy(d + (n & 12), &x2lo, 2);
        const uint32_t x2 = (n & 2) ? x4 >> 16 : x4;
        if (n & 1) d[n & 14] = (uint8_t)x2;
    }
    // a match: n bytes at position d repeat what lies `off` before them
    __device__ __forceinline__ void copy_match(int32_t d, int32_t off, int32_t n)
    {
        if (n <= 16 && off >= n) {  // (text is almost all this)
            write_upto16(d, read16(d - off), n);
            return;
        }
        if (off >= 16) {
            int32_t k = 0;
#pragma unroll 1
            for (; k < n; k += 16) {
                write_upto16(d + k, read16(d - off + k), n - k < 16 ? n - k : 16);
            }
            return;
        }
        int32_t c = 0, dist = off;  // short period: one period first, then twice as far back each step
#pragma unroll 1
        while (c < n) {
            const int32_t m = dist < n - c ? dist : n - c;
#pragma unroll 1
            for (int32_t k = 0; k < m; k += 16) {
                write_upto16(d + c + k, read16(d + c - dist + k), m - k < 16 ? m - k : 16);
            }
            c += m;
            dist += dist;
        }
    }
    // literal bytes from the compressed stream
    __device__ __forceinline__ void copy_literals(int32_t d, const uint8_t* src, int32_t n, const uint8_t* srcEnd)
    {
#pragma unroll 1
        for (int32_t k = 0; k < n; k += 16) {
            const int32_t m = n - k < 16 ? n - k : 16;
            write_upto16(d + k, load_upto16(src + k, m, srcEnd), m);
        }
    }
    // drains the window to the output buffer up to position `to` (uniform; everything below `to` is final): whole 16-byte pieces
    // (positions, not addresses, are 16-aligned: winBase is) by all lanes, a ragged first / last part byte by byte
    __device__ __forceinline__ void flush(int32_t to, bool exactEnd, int lane)
    {
        int32_t from = flushPos;
        if ((from & 15) != 0) {  // ragged start (the window restarted here)
            const int32_t edge = (from + 15) & ~15;
            const int32_t e = edge < to ? edge : to;
            if (from + lane < e) {
                out[from + lane] = win[from + lane - winBase];
            }
            if (e < edge) {
                flushPos = e;
                return;
            }
            from = e;
        }
        const int32_t wholeEnd = to & ~15;
        for (int32_t base = from; base < wholeEnd; base += 1024) {  // (uniform)
            const int32_t p = base + lane * 16;
            if (p < wholeEnd) {
                st16(out + p, *(const u32x4*)(win + (p - winBase)));
            }
        }
        int32_t done = wholeEnd > from ? wholeEnd : from;
        if (exactEnd && done < to) {
            if (done + lane < to) {
                out[done + lane] = win[done + lane - winBase];
            }
            done = to;
        }
        flushPos = done;
    }
    // makes room for a batch behind outPos (uniform): the last SLIDE_KEEP bytes move to the front
    __device__ __forceinline__ void slide(int32_t outPos, int lane)
    {
        if (outPos - winBase + CAP <= WIN) {
            return;
        }
        wave_sync();
        int32_t nb = (outPos - SLIDE_KEEP) & ~15;  // new base (16-aligned position)
        nb = nb > winBase ? nb : winBase;
        const int32_t n = outPos - nb;  // bytes to keep (<= SLIDE_KEEP + 15)
        const int32_t shift = nb - winBase;
        // towards lower addresses, 1 KiB per pass, the passes in ascending order: a pass reads before it writes, and what it overwrites
        // lies below everything later passes read
        for (int32_t base = 0; base < n; base += 1024) {  // (uniform)
            const int32_t i = base + lane * 16;
            u32x4 v0 = u32x4{0, 0, 0, 0};
            if (i < n) v0 = *(const u32x4*)(win + shift + i);
            wave_sync();
            if (i < n) *(u32x4*)(win + i) = v0;
            wave_sync();
        }
        winBase = nb;
    }
};

// One batch of the pipelined executor: what `prepare` derives from 64 records -- and the loads it has started for them.
struct Batch {
    int32_t lit, ml, off;    // this lane's record (zero lengths beyond k)
    int32_t dstLit, srcLit;  // output position of the literal run, its position in the compressed stream
    u32x4 litData;           // the first 16 bytes of the literal run (requested by prepare)
    u32x4 farData;           // the first 16 bytes of a match source that lies in the output buffer only (requested by prepare)
    bool farPre;
    int32_t k;               // (uniform) records in the batch; 0: the first record alone exceeds a batch
    int32_t total, sTotal;   // (uniform) output / compressed bytes of the batch
};

// the records of the next batch: lane i gets record slot + i (i < nb); when the batch ends its chunk, lane nb gets the link
__device__ __forceinline__ uint64_t load_records(const uint64_t* __restrict__ arena, int32_t chunk, int32_t slot, int32_t count, int lane)
{
    int32_t nb = CHUNK_RECS - slot;
    nb = nb < 64 ? nb : 64;
    nb = nb < count ? nb : count;
    const bool link = slot + nb == CHUNK_RECS && nb < 64;
    uint64_t r = 0;
    if (lane < nb || (link && lane == nb)) {
        r = arena[(int64_t)chunk * CHUNK_SLOTS + slot + lane];
    }
    return r;
}

// Same contract as exec_block; `win` = WIN + 16 bytes of LDS owned by this wavefront.  Software pipeline: while batch i is composed in
// the window, the records of batch i + 2 and the literal / far-match bytes of batch i + 1 are on their way.
template <int DBG = 0, int WIN = 4096, int SLIDE_KEEP = 1024>  // DBG: timing aids (output not valid)
__device__ __forceinline__ void exec_block_ring(uint8_t* win, const uint8_t* __restrict__ in, int32_t inLen, uint8_t* out, int32_t outCap,
                                                const uint64_t* __restrict__ arena, int32_t chunk, int32_t count, int lane)
{
    const uint8_t* const inEnd = in + inLen;
    WinIo<WIN, SLIDE_KEEP> io;
    io.win = win;
    io.out = out;
    io.winBase = 0;
    io.flushPos = 0;
    (void)outCap;

    // cursor of `prepare` (one batch ahead of the composing side)
    int32_t pChunk = chunk, pSlot = 0, pCount = count, pOut = 0, pSrc = 0;
    bool pPrevLong = false;  // the batch prepared last is a long record: what lies before the next batch is not in the output buffer yet

    auto prepare = [&](uint64_t r, Batch& b) {
        int32_t nb = CHUNK_RECS - pSlot;
        nb = nb < 64 ? nb : 64;
        nb = nb < pCount ? nb : pCount;
        const bool haveLink = pSlot + nb == CHUNK_RECS && nb < 64;
        const int32_t linkChunk = haveLink ? (int32_t)(uint32_t)shfl_u64(r, nb) : 0;  // (uniform)
        if (lane >= nb) {
            r = 0;
        }
        b.lit = rec_lit(r);
        b.ml = rec_ml(r);
        b.off = rec_off(r);
        const int32_t skip = rec_skip(r);
        const int32_t tot = b.lit + b.ml, adv = skip + b.lit;
        const int32_t oEnd = wave_scan_incl(tot, lane), sEnd = wave_scan_incl(adv, lane);
        b.k = (int32_t)__popcll(__ballot(lane < nb && oEnd <= CAP));  // a prefix: oEnd is monotone
        b.farPre = false;
        b.litData = u32x4{0, 0, 0, 0};
        b.farData = u32x4{0, 0, 0, 0};
        if (b.k == 0) {  // (uniform) the first record alone: moved straight between the global buffers when its turn comes
            b.total = wave_bcast(tot, 0);
            b.sTotal = wave_bcast(adv, 0);
            b.dstLit = pOut;
            b.srcLit = pSrc + wave_bcast(skip, 0);
            b.lit = wave_bcast(b.lit, 0);
            b.ml = wave_bcast(b.ml, 0);
            b.off = wave_bcast(b.off, 0);
        }
        else {
            if (lane >= b.k) {
                b.lit = 0;
                b.ml = 0;
            }
            b.total = wave_bcast(oEnd, b.k - 1);
            b.sTotal = wave_bcast(sEnd, b.k - 1);
            b.dstLit = pOut + oEnd - tot;
            b.srcLit = pSrc + sEnd - (tot - rec_ml(r));
            if (b.lit > 0) {
                b.litData = load_upto16(in + b.srcLit, b.lit, inEnd);
            }
            // a match source that is surely below the window when this batch is composed (the window then starts at or behind
            // pOut - (WIN - CAP)) and was flushed before the PREVIOUS batch began (pOut - WIN + CAP + 16 <= that batch's start - 16)
            const int32_t srcM = b.dstLit + b.lit - b.off;
            if (b.ml > 0 && !pPrevLong && srcM + 16 <= pOut - (WIN - CAP)) {
                b.farData = ld16(out + srcM);
                b.farPre = true;
            }
        }
        pPrevLong = b.k == 0;
        const int32_t consumed = b.k == 0 ? 1 : b.k;
        pOut += b.total;
        pSrc += b.sTotal;
        pSlot += consumed;
        pCount -= consumed;
        if (pSlot == CHUNK_RECS && pCount > 0) {
            // (a batch of exactly 64 records that ends its chunk had no free lane for the link: read it now)
            pChunk = haveLink ? linkChunk : (int32_t)(uint32_t)arena[(int64_t)pChunk * CHUNK_SLOTS + CHUNK_RECS];
            pSlot = 0;
        }
    };

    Batch cur, nxt;
    uint64_t rNext = load_records(arena, pChunk, pSlot, pCount, lane);
    prepare(rNext, cur);
    rNext = pCount > 0 ? load_records(arena, pChunk, pSlot, pCount, lane) : 0;
    int32_t outPos = 0;
    int32_t left = count;
    while (left > 0) {  // (uniform)
        const bool more = pCount > 0;  // (uniform) there is a batch behind `cur`
        if (more) {
            prepare(rNext, nxt);
            rNext = pCount > 0 ? load_records(arena, pChunk, pSlot, pCount, lane) : 0;
        }
        // ---- compose `cur` ----
        if (cur.k == 0) {  // (uniform) one long record
            io.flush(outPos, true, lane);
            wave_sync();
            wave_copy(out + outPos, in + cur.srcLit, cur.lit, lane);
            wave_sync();
            uint8_t* const d = out + outPos + cur.lit;
            if (cur.ml > 0) {
                if (cur.off >= 1024) {
                    wave_copy(d, d - cur.off, cur.ml, lane);
                }
                else if (lane == 0) {
                    copy_match(d, cur.off, cur.ml, d + cur.ml);
                }
            }
            wave_sync();
            outPos += cur.total;
            io.flushPos = outPos;
            io.winBase = outPos & ~15;  // the window restarts here: its first bytes (up to 15) come back from the output buffer
            if (io.winBase + lane < outPos) {
                win[lane] = out[io.winBase + lane];
            }
            wave_sync();
            left -= 1;
        }
        else {
            io.slide(outPos, lane);
            const int32_t dstM = cur.dstLit + cur.lit;
            const int32_t srcM = dstM - cur.off;
            const int32_t span = cur.ml < cur.off ? cur.ml : cur.off;  // source bytes that are not the match's own output
            bool pending = cur.ml > 0 && DBG != 1 && DBG != 4;
            // literal runs (the first 16 bytes are here already) and the far matches whose bytes are here as well
            if (cur.lit > 0 && DBG != 2 && DBG != 4) {
                io.write_upto16(cur.dstLit, cur.litData, cur.lit < 16 ? cur.lit : 16);
                if (cur.lit > 16) {
                    io.copy_literals(cur.dstLit + 16, in + cur.srcLit + 16, cur.lit - 16, inEnd);
                }
            }
            if (pending && cur.farPre && cur.ml <= 16) {
                io.write_upto16(dstM, cur.farData, cur.ml);
                pending = false;
            }
            // a match whose source reaches into this batch's output waits for exactly the lanes a .. b-1 that produce it (outputs are
            // contiguous and ordered over the lanes: two binary searches by lane shuffles), as far as they are pending themselves
            unsigned long long dep = 0;
            const bool inBatch = pending && srcM + span > outPos;
            const unsigned long long producers = __ballot(pending);  // every match still to be written may be somebody's source
            if (__ballot(inBatch) != 0 && DBG != 6) {  // (uniform)
                const int32_t myStart = cur.dstLit, myEnd = dstM + cur.ml;
                int32_t a = 0, bnd = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const int32_t e = __shfl(myEnd, a + step - 1);
                    const int32_t st = __shfl(myStart, bnd + step - 1);
                    a += e <= srcM ? step : 0;            // lanes that end at or before the source's start
                    bnd += st < srcM + span ? step : 0;   // lanes that start before the source's end
                }
                bnd = bnd < lane ? bnd : lane;
                if (inBatch && a < bnd) {
                    dep = producers & ((bnd >= 64 ? ~0ull : ((1ull << bnd) - 1)) & ~((1ull << a) - 1));
                }
            }
            if (DBG == 6 && inBatch) pending = false;
            wave_sync();
            for (;;) {  // (uniform) round 1: everything that waits for nothing; the first pending lane is always ready
                const unsigned long long pm = __ballot(pending);
                if (pm == 0) {
                    break;
                }
                if (pending && (dep & pm) == 0) {
                    io.copy_match(dstM, cur.off, cur.ml);
                    pending = false;
                }
                wave_sync();
            }
            outPos += cur.total;
            if (DBG != 3 && DBG != 4) {
                io.flush(outPos, false, lane);
            }
            left -= cur.k;
        }
        if (more) {
            cur = nxt;
        }
    }
    wave_sync();
    io.flush(outPos, true, lane);
}

}  // namespace sx
}  // namespace achip
