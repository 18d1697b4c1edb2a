// achip_seqexec.h -- what the two passes of the record-based LZ77 decoders share (lz4_decompress_v7.hip, snappy_decompress_v5.hip):
// the 8-byte RECORD a parser writes per piece of work and the chunked arena the records live in, wave helpers, and the parser's view of
// its compressed stream (LaneFeed).  The executor is achip_seqexec2.h.
//
//   parse    a lane per block walks the token grammar (serial by nature) and writes one record per piece
//            {literal length <= 16, match length <= 16, offset, header bytes skipped} into the arena -- no byte is copied;
//   execute  a WAVEFRONT per block runs 64 records at a time.
// With a wavefront per block only a few thousand blocks are open at a time (the lane-per-block decoders keep 131072+ open: 8x more memory
// traffic than bytes decoded on text), input and record streams are read coalesced, and the output is written in stream order.
//
// Cross-lane operations are only used in wave-uniform control flow and lanes exchange data through memory only across wave_sync():
// the kernels built on this header run unchanged under tools/hostemu (fibers) on a CPU.
#pragma once
#include "achip_device.h"

namespace achip {

namespace sx {

// ---- records --------------------------------------------------------------------------------------------------------------------
// bits  0..16  literal length   (the parsers emit pieces: <= 16)
// bits 17..33  match length     (likewise <= 16; 0 = none)
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
        // (an empty payload has no byte of its own to anchor the granule loads at: `in` may be the first byte BEHIND the item, and on a
        // 32-byte boundary the piece at `in` then lies outside everything the caller handed over -- such a stream reads `anywhere`)
        inAligned = inLimit > 0 ? in - inBase : (const uint8_t*)anywhere;
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

}  // namespace sx
}  // namespace achip
