// achip_device.h -- device-side building blocks shared by the gfx950 codec kernels.
//
// Execution shape used by the LZ77 decoders: a wavefront (64 lanes) is split into
// 64/GS "groups" of GS consecutive lanes; each group owns one block of the batch.
// Everything that is serial in the format (token grammar) is computed redundantly
// by all GS lanes of the group (same addresses => one memory transaction), and
// everything that moves bytes is spread over the GS lanes 16 bytes per lane, so a
// group step touches GS*16 contiguous bytes of HBM.  Groups of one wave diverge only
// on rare paths (length escapes, tails, overlapping matches).
#pragma once
#ifndef ACHIP_WAVES_PER_EU  // cap a kernel's registers for at least `lo` wavefronts per SIMD (the CPU emulator's shim defines it away)
#define ACHIP_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aircompressor_hip.h"

namespace achip {

struct BatchArgs {
    const uint8_t* __restrict__ srcBase;
    const int64_t* __restrict__ srcOff;
    const int32_t* __restrict__ srcLen;
    uint8_t* __restrict__ dstBase;
    const int64_t* __restrict__ dstOff;
    const int32_t* __restrict__ dstCap;
    int32_t* __restrict__ outLen;
    int32_t* __restrict__ status;
    int64_t* __restrict__ errOffset;
    int32_t nBlocks;
    int32_t ringPad;  // LDS padding between the ring pairs of consecutive blocks (decoders)
    const int32_t* nBlocksDev = nullptr;  // when set: the number of blocks is this device word (<= nBlocks, which then sizes the launch): a
                                // batch assembled on the device (the chunk list of the framed readers)
    const int32_t* only = nullptr;        // when set (ring decoders): decode block i only if only[i] != 0 -- the blocks the two-pass decoders
                                // hand over (lz4_decompress_v7.hip)
    const int32_t* onlyStats = nullptr;   // ... and only if these probe statistics picked the two-pass decoder (auto mode; else null)
    int32_t onlyShortLimit = 12;
    // ring decoders, batches assembled on the device: run only if countLo <= the batch's block count < countHi (the stream readers launch two
    // lane-group sizes and the count, known on the device only, picks one: few large blocks want more lanes each)
    int32_t countLo = 0, countHi = 0x7FFFFFFF;
};

__device__ __forceinline__ int32_t batch_count(const BatchArgs& a) { return a.nBlocksDev != nullptr ? *a.nBlocksDev : a.nBlocks; }

__device__ __forceinline__ constexpr int32_t mk_status(int cls, int detail) { return -(cls + 16 * detail); }

// LZ4 decoder choice of the auto mode: a batch is "mixed" when in more than a quarter of its groups of 16 consecutive
// blocks (= the blocks one wavefront of the ring decoder works on) the compressed sizes differ by more than 2x
__device__ __forceinline__ bool lz4_batch_is_mixed(int32_t mixedGroups, int32_t nBlocks) { return (int64_t)mixedGroups * 4 > (nBlocks + 15) / 16; }

// The decoders' choice.  stats: [0] mixed groups, [1] sequences parsed from the heads of 1024 sampled blocks, [2] the bytes they produce,
// in units of 4, [4] / [5] sampled blocks of short sequences / sampled blocks (0: not counted), [3] != 0: the caller has record scratch -- the two-pass decoder (parse to records + a wavefront per block) is
// a candidate.  Mixed batches (long copies next to short ones) and short-sequence batches (text: 9 .. 40 bytes per sequence at a block's
// head; the long-copy sets: >= 100) go to it, everything else -- and every batch of a caller without record scratch -- to the rings.
// (Until round 4 there were two more candidates, lane-per-block decoders; the two-pass decoder beat both from 16384 blocks up.)
constexpr int LZ4_PICK_RINGS = 0, LZ4_PICK_TWOPASS = 3;
__device__ __forceinline__ int lz4_pick(const int32_t* stats, int32_t nBlocks, int32_t shortLimit = 12)
{
    // Short-sequence data, decided block by block where the probe says so (round 4): [4] of the [5] sampled blocks have short sequences.  The pooled
    // mean of rounds 2-3 -- all sampled bytes / all sampled sequences -- let a handful of long-run blocks (thousands of bytes per sequence) outvote a
    // batch of text: a batch of text and long copies side by side went to the rings at 227 GiB/s where the two passes make 503.  A third of the
    // blocks short is where the two passes win (text: 515 against ~200 for the rings; long copies: ~450 against 1 900).
    const bool pooledShort = stats[1] > 0 && (int64_t)stats[2] < (int64_t)shortLimit * (int64_t)stats[1];
    const bool isShort = stats[5] > 0 ? (int64_t)stats[4] * 3 > (int64_t)stats[5] : pooledShort;
    return (stats[3] != 0 && (lz4_batch_is_mixed(stats[0], nBlocks) || isShort)) ? LZ4_PICK_TWOPASS : LZ4_PICK_RINGS;
}
// Snappy: the sample counts elements (a literal run or a copy -- half an LZ4 sequence)
__device__ __forceinline__ int snappy_pick(const int32_t* stats, int32_t nBlocks) { return lz4_pick(stats, nBlocks, 6); }

// ---- unaligned little-endian accessors (global memory; gfx950 runs in unaligned-access mode) ----
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x4 ld16(const uint8_t* p)
{
    u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
// 16 aligned bytes of GLOBAL memory: the address-space cast keeps the load a global_load where the compiler cannot prove the pointer's
// provenance (a flat_load also counts against the LDS counter: every wait for an LDS read would wait for it)
__device__ __forceinline__ u32x4 ld16_global(const uint8_t* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const u32x4 __attribute__((address_space(1))) * GlobalPtr;
    return *(GlobalPtr)(p);
#else
    return *(const u32x4*)p;
#endif
}
__device__ __forceinline__ void st16(uint8_t* p, u32x4 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ uint64_t ld8(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void st8(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ uint32_t ld4(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ void st4(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ uint32_t ld2(const uint8_t* p)
{
    uint16_t v;
    __builtin_memcpy(&v, p, 2);
    return v;
}
__device__ __forceinline__ void st2(uint8_t* p, uint32_t v)
{
    uint16_t w = (uint16_t)v;
    __builtin_memcpy(p, &w, 2);
}

// dynamic LDS of a kernel; tools/hostemu (plain clang++, kernels run one lane at a time) substitutes a host arena
#if defined(__HIPCC__)
#define ACHIP_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]
#else
#define ACHIP_DYNAMIC_LDS(name) uint8_t* name = hostemu_dynamic_lds
#endif

// (hi:lo) >> 8*s for s in 0..3 -- v_alignbyte_b32 on the device, plain C when the kernel sources are built for the host (tools/hostemu)
__device__ __forceinline__ uint32_t alignbyte_u32(uint32_t hi, uint32_t lo, uint32_t s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (s & 3)));
#endif
}

// Compiler-level ordering point between a store phase and a load phase whose addresses
// may be produced by OTHER lanes of the same wave.  The hardware issues a wave's vector
// memory instructions to the CU's L1/TA in program order, so keeping the compiler from
// reordering across this point is sufficient inside one wave.
#if !defined(__HIPCC__) && defined(HOSTEMU_ORDER_IS_RENDEZVOUS)
// (tools/hostemu, translation units that hold only kernels running one item per WAVEFRONT in wave-uniform control flow -- the
// wavefront-per-item readers of the containers, the one-kernel Zstd decoder: the hand-over point is a rendezvous of the wave there)
#define wave_mem_order() hostemu::order_point(__FILE__, __LINE__)
#elif !defined(__HIPCC__) && defined(HOSTEMU_ORDER_IS_SOFT)
// (tools/hostemu, any mix of kernels: the hand-over point is a pause until no lane of the wave can run further -- a barrier for code in
// wave-uniform control flow, harmless for lanes that go their own ways)
#define wave_mem_order() hostemu::soft_order_point(__FILE__, __LINE__)
#else
__device__ __forceinline__ void wave_mem_order() { asm volatile("" ::: "memory"); }
#endif

// "This value is in its register from here on": a load in flight into `v` is waited for at this point, in straight-line code.  For values that
// are loaded in front of a loop and used inside it -- LLVM's wait-count insertion otherwise carries "maybe still in flight" into the loop and waits
// for it in EVERY trip, with a count that also drains the stores the trip before issued (profiles/r06_notes.md, the Zstd sequence stage).
template <class T>
__device__ __forceinline__ void settle(T& v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#else
    asm volatile("" : "+r"(v));
#endif
}

// The same ordering point for kernels whose lanes COOPERATE on one block: everything the lanes of this wave stored before it (LDS or
// global) is visible to every lane after it.  On the device a compiler barrier is all it takes (a wavefront's memory operations are
// performed in program order; wavefront-scope fences are no-ops on gfx950).  Must be called in wave-uniform control flow: tools/hostemu
// runs the lanes as fibers and makes this a rendezvous, which is how the cooperative kernels are tested on a CPU.
#if defined(__HIPCC__)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
#else
#define wave_sync() hostemu::wave_sync(__FILE__, __LINE__)
#endif

// The Zstd decode pipeline's multi-block stages ask their caller for scratch only when a batch holds multi-block frames (the request
// comes after a stream synchronisation: `get` may allocate; nullptr = none, the frames then take the slow path).  passBlocks: 128 KiB
// blocks per pass through the stages.
// The same arrangement for other launchers that learn what they need only on the device (the Hadoop block streams' chunk count): more
// scratch on request, after a stream synchronisation.
struct AuxScratch {
    void* (*get)(void* user, int64_t bytes);
    void* user;
};
struct ZstdMbProvider {
    void* (*get)(void* user, int64_t bytes);
    void* user;
    int32_t passBlocks;
};

// Kernels that give an item to a QUAD of lanes (the Zstd pipeline's sequence stage): the value of lane K of the caller's quad (a DPP
// quad_perm broadcast, no LDS traffic), and the ordering point for data that goes between the lanes of a quad through LDS.  Both only
// need quad-uniform control flow; tools/hostemu makes them rendezvous of the quad.
#if defined(__HIPCC__)
template <int K>
__device__ __forceinline__ int32_t quad_bcast(int32_t v)
{
    return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, false);  // quad_perm:[K,K,K,K]
}
__device__ __forceinline__ void quad_sync() { asm volatile("" ::: "memory"); }
#else
template <int K>
inline int32_t quad_bcast(int32_t v)
{
    return hostemu::quad_from(v, K, __FILE__, __LINE__);
}
#define quad_sync() hostemu::quad_sync(__FILE__, __LINE__)
#endif

// the value lane k of the caller's GS-lane group holds (laneBase: the group's first lane within the wavefront).  A macro, so that the device
// code is the plain __shfl expression; tools/hostemu makes it a rendezvous of the group only
#if defined(__HIPCC__)
#define ACHIP_GROUP_SHFL(GS, v, laneBase, k) __shfl((v), (laneBase) + (k))
#else
#define ACHIP_GROUP_SHFL(GS, v, laneBase, k) hostemu::group_from((v), (GS), (k), __FILE__, __LINE__)
#endif

// ---- group copy: n bytes, src and dst ranges do not overlap (or src+n <= dst) ----
// Lane g of a GS-lane group moves bytes [16g,16g+16) of every GS*16-byte step; the
// ragged tail (n mod 16) is moved one byte per lane.  Exact: never reads or writes
// outside [src,src+n) / [dst,dst+n).
template <int GS>
__device__ __forceinline__ void group_copy(uint8_t* dst, const uint8_t* src, int32_t n, int g)
{
    const int32_t full = n & ~15;
    for (int32_t base = g * 16; base < full; base += GS * 16) {
        st16(dst + base, ld16(src + base));
    }
    if constexpr (GS >= 4) {
        for (int32_t k = full + g; k < n; k += GS) {
            dst[k] = src[k];
        }
    }
    else {
        const int32_t r = n & 15;
        if (r != 0 && g == ((n >> 4) % GS)) {
            int32_t k = full;
            if (r & 8) {
                st8(dst + k, ld8(src + k));
                k += 8;
            }
            if (r & 4) {
                st4(dst + k, ld4(src + k));
                k += 4;
            }
            if (r & 2) {
                st2(dst + k, ld2(src + k));
                k += 2;
            }
            if (r & 1) {
                dst[k] = src[k];
            }
        }
    }
}

// ---- group match copy: LZ77 back-reference inside the output buffer ----
// out[pos+k] = out[pos-offset+k] for k in [0,len), byte-sequential semantics (the
// source may overlap the destination when offset < len).  Overlap is resolved without
// serialising on bytes: a prefix of the match that lies within one period is a plain
// non-overlapping copy, and once 2^r periods exist the next 2^r periods can be copied
// from them at distance offset*2^r.  For offset < 16 the first 256 bytes are produced
// byte-per-lane from the period (source index = k mod offset), all reading data that
// existed before the match started.
template <int GS>
__device__ __forceinline__ void group_match_copy(uint8_t* out, int32_t pos, int32_t offset, int32_t len, int g)
{
    wave_mem_order();
    uint8_t* dst = out + pos;
    int32_t copied = 0;
    int32_t d = offset;
    if (offset < 16 && len > offset) {
        const int32_t head = len < 256 ? len : 256;
        const uint8_t* period = dst - offset;
        const uint32_t inv = (65536u + (uint32_t)offset - 1u) / (uint32_t)offset;  // exact k/offset for k < 4096
        for (int32_t k = g; k < head; k += GS) {
            const uint32_t q = ((uint32_t)k * inv) >> 16;
            dst[k] = period[k - (int32_t)q * offset];
        }
        copied = head;
        d = (256 / offset) * offset;
        wave_mem_order();
    }
    while (copied < len) {
        const int32_t rem = len - copied;
        const int32_t n = rem < d ? rem : d;
        group_copy<GS>(dst + copied, dst + copied - d, n, g);
        copied += n;
        if (copied < len) {
            if (d < (1 << 20)) {
                d += d;
            }
            wave_mem_order();
        }
    }
}

// "this value is the same in every lane."  State that is wave-uniform by construction (a position, a length, a count) but was loaded from memory
// or came through a shuffle is a vector register to the compiler, and one such value in a position turns every branch and every mask operation
// behind it into vector code under exec masks.  uni() hands the compiler the scalar.
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

// The opposite direction (round 6): "keep this wave-uniform value in a VECTOR register, and do what is computed from it with vector instructions".  A CU has
// one scalar unit for its four SIMDs; the window encoders' replay loops (lz4_compress_mw.h, snappy_compress_mw.h) had become 70 % scalar instructions and
// were bound by that unit while the vector units idled (profiles/r06_notes.md section 4).  Straight-line arithmetic on a sequence's positions and lengths costs
// the same instruction count on either side -- and none of the scalar side's s_cselect / s_and pairs and branches per condition.
__device__ __forceinline__ int32_t vec(int32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(v));
#endif
    return v;
}
__device__ __forceinline__ uint32_t vec(uint32_t v) { return (uint32_t)vec((int32_t)v); }

// ---- wave-wide match-length count (one wavefront per block) ----
// Number of equal bytes of in[a..] vs in[b..] with a < limit, b < a: the `count` routines of the Java
// encoders (M/lz4/Lz4RawCompressor.java:240-267, M/snappy/SnappyRawCompressor.java:235-266) both return
// the exact common-prefix length capped at the limit.  64 lanes x 8 bytes per step; wave-uniform result.
__device__ __forceinline__ int32_t wave_count(const uint8_t* __restrict__ in, int32_t a, int32_t b, int32_t limit, int lane)
{
    int32_t total = 0;
    for (;;) {
        const int32_t remaining = limit - a;  // bytes still comparable
        const int32_t o = lane * 8;
        int32_t chunk = remaining - o;
        chunk = chunk > 8 ? 8 : chunk;
        int32_t eq = 0;
        if (chunk == 8) {
            const uint64_t diff = ld8(in + a + o) ^ ld8(in + b + o);
            eq = diff == 0 ? 8 : (__builtin_ctzll(diff) >> 3);
        }
        else if (chunk > 0) {
            while (eq < chunk && in[a + o + eq] == in[b + o + eq]) {
                eq++;
            }
        }
        const bool stop = chunk < 8 || eq < 8;
        const unsigned long long m = __ballot(stop);
        if (m != 0) {
            const int first = __builtin_ctzll(m);
            const int32_t e = __builtin_amdgcn_readlane(eq, first);  // (a scalar: the callers' positions stay wave-uniform for the compiler, too)
            return total + first * 8 + e;
        }
        total += 512;
        a += 512;
        b += 512;
    }
}

// ---- wave-wide fill and XXH64 (one wavefront per item) ----
__device__ __forceinline__ void wave_fill(uint8_t* dst, int32_t value, int32_t n, int lane)
{
    const uint32_t v4 = (uint32_t)value * 0x01010101u;
    const int32_t full = n & ~15;
    u32x4 v = {v4, v4, v4, v4};
    for (int32_t base = lane * 16; base < full; base += 64 * 16) {
        st16(dst + base, v);
    }
    for (int32_t k = full + lane; k < n; k += 64) {
        dst[k] = (uint8_t)value;
    }
}

// XXH64 (seed 0) of out[0..len) -- M/zstd/XxHash64.java:182-291.  Lanes 0-3 own the four accumulators.
__device__ inline uint64_t wave_xxh64(const uint8_t* p, int32_t len, int lane)
{
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto mix = [&](uint64_t cur, uint64_t v) { return rotl(cur + v * P2, 31) * P1; };
    uint64_t hash;
    if (len >= 32) {
        uint64_t v = lane == 0 ? P1 + P2 : (lane == 1 ? P2 : (lane == 2 ? 0 : (0 - P1)));
        const int32_t stripes = len >> 5;
        if (lane < 4) {
            const uint8_t* q = p + lane * 8;
            for (int32_t s = 0; s < stripes; s++) {
                v = mix(v, ld8(q + (int64_t)s * 32));
            }
        }
        const uint64_t v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
        hash = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        hash = (hash ^ mix(0, v1)) * P1 + P4;
        hash = (hash ^ mix(0, v2)) * P1 + P4;
        hash = (hash ^ mix(0, v3)) * P1 + P4;
        hash = (hash ^ mix(0, v4)) * P1 + P4;
    }
    else {
        hash = P5;
    }
    hash += (uint64_t)len;
    int32_t index = len & ~31;
    while (index <= len - 8) {
        hash = rotl(hash ^ mix(0, ld8(p + index)), 27) * P1 + P4;
        index += 8;
    }
    if (index <= len - 4) {
        hash = rotl(hash ^ ((uint64_t)ld4(p + index) * P1), 23) * P2 + P3;
        index += 4;
    }
    while (index < len) {
        hash = rotl(hash ^ ((uint64_t)p[index] * P5), 11) * P1;
        index++;
    }
    hash ^= hash >> 33;
    hash *= P2;
    hash ^= hash >> 29;
    hash *= P3;
    hash ^= hash >> 32;
    return hash;
}

// first lane index of this lane's group, lane index inside the group
template <int GS>
__device__ __forceinline__ int group_lane()
{
    return (int)(threadIdx.x & (GS - 1));
}

}  // namespace achip
