// achip_lanecopy.h -- the machinery of the lane-per-block decoders (lz4_decompress_v5.hip, snappy_decompress_v3.hip): every
// lane of a wavefront owns a block; its compressed stream is read through an LDS window (LaneInput), its output is
// written straight to the global buffer by wavefront-wide copy steps (copy_step) in which all loads are issued before
// the stores.  See lz4_decompress_v5.hip for the design notes and profiles/r01_notes.md for the measurements.
#pragma once
#include "achip_device.h"

namespace achip {
namespace sp {

// 16-byte load that does not ask the L2 to keep the line: back-reference sources are touched once, and every line they
// would park in the L2 pushes out a half-written output line of some other block (262144 blocks are open at once)
template <bool NT>
__device__ __forceinline__ u32x4 ld16_once(const uint8_t* p)
{
    // (the address need not be 16-byte aligned: gfx950 runs in unaligned-access mode and this is one load instruction)
    if constexpr (NT) {
        return __builtin_nontemporal_load((const u32x4*)p);
    }
    else {
        return ld16(p);
    }
}

// The lane's window on its compressed stream: an LDS ring column fed with aligned 32-byte pieces (two 16-byte loads of one
// half line), one piece requested ahead.  With 262144 streams open a cache line does not survive in the L2 until its
// stream comes back for the next piece (measured with 16-byte pieces: 69 GB fetched for 8.7 GB of input); larger pieces
// fetch a line fewer times (64-byte pieces cost 18 more registers and a wave per SIMD: slower overall).
template <int IN_DW>
struct LaneInput {
    static constexpr int IN_BYTES = IN_DW * 4;
    static constexpr int PIECE = 32;
    static_assert((IN_DW & (IN_DW - 1)) == 0 && IN_DW >= 16, "ring size");
    uint32_t* inR;  // dword d of this lane's ring at inR[(d & (IN_DW-1)) * 64]
    const uint8_t* inAligned;
    int32_t inBase;
    int32_t inEndV;
    int32_t inLoadedV;     // virtual [.., inLoadedV) is in the ring (as far back as the ring reaches)
    u32x4 pendingA, pendingB;  // the piece at inLoadedV

    __device__ __forceinline__ void init(uint32_t* lds, const uint8_t* in, int32_t inLimit)
    {
        inR = lds;
        inBase = (int32_t)((uintptr_t)in & (PIECE - 1));
        inAligned = in - inBase;
        inEndV = inLimit + inBase;
        inLoadedV = 0;
        pendingA = fetch_granule(0);
        pendingB = fetch_granule(16);
    }
    __device__ __forceinline__ u32x4 fetch_granule(int32_t v) const
    {
        u32x4 d = {0, 0, 0, 0};
        if (v >= inBase && v + 16 <= inEndV) {
            d = *(const u32x4*)(inAligned + v);
        }
        else if (v + 16 > inBase && v < inEndV) {  // first / last granule: byte-guarded (cold)
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
            for (int i = 0; i < 16; i++) {
                const int32_t p = v + i;
                if (p >= inBase && p < inEndV) {
                    w[i >> 2] |= (uint32_t)inAligned[p] << (8 * (i & 3));
                }
            }
            d = u32x4{w[0], w[1], w[2], w[3]};
        }
        return d;
    }
    // make [pos, pos + need) resident (need <= IN_BYTES - PIECE; bytes past the end read as 0).  The stream is only read
    // forwards, so a jump over literal bytes restarts the ring at the piece of pos.
    __device__ __forceinline__ void ensure_input(int32_t pos, int32_t need)
    {
        const int32_t v = pos + inBase;
        if (v >= inLoadedV + PIECE) {
            inLoadedV = v & ~(PIECE - 1);
            pendingA = fetch_granule(inLoadedV);
            pendingB = fetch_granule(inLoadedV + 16);
        }
        while (v + need > inLoadedV && inLoadedV < inEndV) {
            const int32_t d = inLoadedV >> 2;
            inR[((d + 0) & (IN_DW - 1)) * 64] = pendingA.x;
            inR[((d + 1) & (IN_DW - 1)) * 64] = pendingA.y;
            inR[((d + 2) & (IN_DW - 1)) * 64] = pendingA.z;
            inR[((d + 3) & (IN_DW - 1)) * 64] = pendingA.w;
            inR[((d + 4) & (IN_DW - 1)) * 64] = pendingB.x;
            inR[((d + 5) & (IN_DW - 1)) * 64] = pendingB.y;
            inR[((d + 6) & (IN_DW - 1)) * 64] = pendingB.z;
            inR[((d + 7) & (IN_DW - 1)) * 64] = pendingB.w;
            inLoadedV += PIECE;
            pendingA = fetch_granule(inLoadedV);
            pendingB = fetch_granule(inLoadedV + 16);
        }
        wave_mem_order();
    }
    __device__ __forceinline__ uint64_t in_u64(int32_t pos) const
    {
        const int32_t v = pos + inBase;
        const int32_t d = v >> 2;
        const uint32_t w0 = inR[((d + 0) & (IN_DW - 1)) * 64], w1 = inR[((d + 1) & (IN_DW - 1)) * 64], w2 = inR[((d + 2) & (IN_DW - 1)) * 64];
        const uint32_t s = (uint32_t)(v & 3);
        return ((uint64_t)alignbyte_u32(w2, w1, s) << 32) | alignbyte_u32(w1, w0, s);
    }
    __device__ __forceinline__ uint32_t in_u8(int32_t pos) const
    {
        const int32_t v = pos + inBase;
        return (inR[((v >> 2) & (IN_DW - 1)) * 64] >> (8 * (v & 3))) & 0xFF;
    }
    // 16 input bytes at pos (resident: ensure_input(pos, 20))
    __device__ __forceinline__ u32x4 in_u128(int32_t pos) const
    {
        const int32_t v = pos + inBase;
        const int32_t d = v >> 2;
        const uint32_t r0 = inR[((d + 0) & (IN_DW - 1)) * 64], r1 = inR[((d + 1) & (IN_DW - 1)) * 64], r2 = inR[((d + 2) & (IN_DW - 1)) * 64],
                       r3 = inR[((d + 3) & (IN_DW - 1)) * 64], r4 = inR[((d + 4) & (IN_DW - 1)) * 64];
        const uint32_t s = (uint32_t)(v & 3);
        return u32x4{alignbyte_u32(r1, r0, s), alignbyte_u32(r2, r1, s), alignbyte_u32(r3, r2, s), alignbyte_u32(r4, r3, s)};
    }
};

// inclusive prefix sum over aligned segments of SEG (8 or 16) lanes
template <int SEG>
__device__ __forceinline__ int32_t seg_scan(int32_t x, int j)
{
    int32_t t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x += (SEG == 16 || j >= 1) ? t : 0;
    t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += (SEG == 16 || j >= 2) ? t : 0;
    t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += (SEG == 16 || j >= 4) ? t : 0;
    if constexpr (SEG == 16) {
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    }
    return x;
}
// inclusive prefix sum over the wavefront
__device__ __forceinline__ int32_t wave_scan(int32_t x, int lane)
{
    x = seg_scan<16>(x, lane & 15);
    const int32_t r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
    return x + (lane >= 16 ? r0 : 0) + (lane >= 32 ? r1 : 0) + (lane >= 48 ? r2 : 0);
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int srcLane)
{
    return ((uint64_t)(uint32_t)__shfl((int32_t)(v >> 32), srcLane) << 32) | (uint32_t)__shfl((int32_t)v, srcLane);
}

constexpr int HEAD = 32;   // bytes of a copy its own lane moves per trip
constexpr int BIG = 1024;  // ... and with more than this, moved by the whole wavefront, one copy after the other
constexpr int LONG = 128;  // a copy with more than this left is cut into 16-byte chunks that are dealt out to all lanes

struct CopyScratch {  // LDS, per wavefront: the copies of the current step, for the chunk loop
    uint32_t pre[64];    // running chunk count (inclusive) over the lanes
    uint32_t c0[64];     // chunks of the lane's first copy
    uint32_t n[2][64];
    uint64_t dst[2][64], src[2][64];
};

struct HeadRegs {
    u32x4 A, B;
};

// first <= HEAD bytes of a copy: loads.  `srcEnd` bounds what may be read: a short run is fetched with one 16-byte load
// when that stays inside the buffer (the bytes past the run are not used), byte by byte otherwise (cold).
template <bool NT>
__device__ __forceinline__ void head_load(HeadRegs& r, const uint8_t* src, int32_t m, const uint8_t* srcEnd)
{
    r.A = u32x4{0, 0, 0, 0};
    r.B = r.A;
    if (m > 0) {
        if (src + 16 <= srcEnd) {
            r.A = ld16_once<NT>(src);
        }
        else {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
            for (int i = 0; i < m && i < 16; i++) {
                w[i >> 2] |= (uint32_t)src[i] << (8 * (i & 3));
            }
            r.A = u32x4{w[0], w[1], w[2], w[3]};
        }
        if (m >= 16) {
            r.B = ld16_once<NT>(src + m - 16);
        }
    }
}
// Stores.  A run shorter than 16 bytes is written with ONE 16-byte store when that stays inside the block's output
// capacity `dstEnd`: the bytes behind the run are the lane's own next output positions and are written again, in program
// order, by its later copies (the Java fast path overshoots the same way, within the buffer, 8 bytes at a time
// M/lz4/Lz4RawDecompressor.java:98-104,174-187).  Close to the end of the capacity: exact, 8/4/2/1.
__device__ __forceinline__ void head_store(const HeadRegs& r, uint8_t* dst, int32_t m, const uint8_t* dstEnd)
{
    if (m >= 16) {
        st16(dst, r.A);
        st16(dst + m - 16, r.B);
    }
    else if (m > 0 && dst + 16 <= dstEnd) {
        st16(dst, r.A);
    }
    else if (m > 0) {
        const uint64_t lo = ((uint64_t)r.A.y << 32) | r.A.x, hi = ((uint64_t)r.A.w << 32) | r.A.z;
        if (m & 8) st8(dst, lo);
        const uint64_t x8 = (m & 8) ? hi : lo;
        if (m & 4) st4(dst + (m & 8), (uint32_t)x8);
        const uint32_t x4 = (m & 4) ? (uint32_t)(x8 >> 32) : (uint32_t)x8;
        if (m & 2) st2(dst + (m & 12), x4);
        const uint32_t x2 = (m & 2) ? x4 >> 16 : x4;
        if (m & 1) dst[m & 14] = (uint8_t)x2;
    }
}

// One wavefront-wide copy step: every lane has up to two copies (n0 bytes src0 -> dst0, then n1 bytes src1 -> dst1; a
// length of 0 = none).  Within a lane the ranges do not overlap and every source byte is final before the step.  Exact.
// `have0`: the first 16 bytes of copy 0 are in h0.A already (a literal run of <= 16 bytes comes from the LDS window).
// `end1` is the end of the output capacity (copy 1 reads the output buffer; both copies write it).
template <bool TWO, bool NT = false>
__device__ __forceinline__ void copy_step(CopyScratch& S, int lane, HeadRegs h0, bool have0, uint8_t* dst0, const uint8_t* src0, int32_t n0,
                                          const uint8_t* end0, uint8_t* dst1, const uint8_t* src1, int32_t n1, const uint8_t* end1)
{
    HeadRegs h1;
    const int32_t m0 = n0 < HEAD ? n0 : HEAD, m1 = n1 < HEAD ? n1 : HEAD;
    if constexpr (TWO) {
        if (!have0) {
            head_load<false>(h0, src0, m0, end0);
        }
    }
    head_load<NT>(h1, src1, m1, end1);
    const bool longer = (TWO && n0 > HEAD) || n1 > HEAD;
    const bool anyLonger = __ballot(longer) != 0;
    if (anyLonger) {  // (uniform)
        // more than BIG bytes: a plain memcpy by the whole wavefront, bases in scalar registers, 4 KiB per round
        const bool big0 = TWO && n0 > BIG, big1 = n1 > BIG;
#pragma unroll
        for (int which = TWO ? 0 : 1; which < 2; which++) {
            for (unsigned long long m = __ballot(which ? big1 : big0); m != 0; m &= m - 1) {
                const int l = __builtin_ctzll(m);
                const uint64_t sv = (uint64_t)(uintptr_t)(which ? src1 : src0), dv = (uint64_t)(uintptr_t)(which ? dst1 : dst0);
                const uint8_t* const s = (const uint8_t*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(sv >> 32), l) << 32) |
                                                                     (uint32_t)__builtin_amdgcn_readlane((int32_t)sv, l));
                uint8_t* const d = (uint8_t*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(dv >> 32), l) << 32) |
                                                         (uint32_t)__builtin_amdgcn_readlane((int32_t)dv, l));
                const int32_t len = __builtin_amdgcn_readlane(which ? n1 : n0, l);
                for (int32_t base = HEAD + 16 * lane; base < len; base += 4096) {
                    u32x4 v[4];
                    int32_t p[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        p[t] = base + 1024 * t;
                        p[t] = p[t] + 16 > len ? len - 16 : p[t];  // (the last piece ends at the end; pieces past it repeat it)
                        v[t] = ld16_once<NT>(s + p[t]);
                    }
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        st16(d + p[t], v[t]);
                    }
                }
            }
        }
        // the rest: publish the copies for the chunk loop below
        const int32_t c0 = (TWO && n0 > HEAD && !big0) ? (n0 - HEAD + 15) >> 4 : 0;
        const int32_t c1 = (n1 > HEAD && !big1) ? (n1 - HEAD + 15) >> 4 : 0;
        S.pre[lane] = (uint32_t)wave_scan(c0 + c1, lane);
        S.c0[lane] = (uint32_t)c0;
        S.n[0][lane] = (uint32_t)n0;
        S.n[1][lane] = (uint32_t)n1;
        S.dst[0][lane] = (uint64_t)(uintptr_t)dst0;
        S.src[0][lane] = (uint64_t)(uintptr_t)src0;
        S.dst[1][lane] = (uint64_t)(uintptr_t)dst1;
        S.src[1][lane] = (uint64_t)(uintptr_t)src1;
    }
    if (!anyLonger) {
        if constexpr (TWO) {
            head_store(h0, dst0, m0, end1);
        }
        head_store(h1, dst1, m1, end1);
        wave_mem_order();
        return;
    }
    wave_mem_order();
    const int32_t total = (int32_t)S.pre[63];
    // first batch of chunks is loaded before the heads are stored: everything of a short step is in flight together
    bool headsStored = false;
    constexpr int T = 2;  // chunks per lane in flight
    for (int32_t c0 = 0; c0 < total; c0 += 64 * T) {
        u32x4 v[T];
        uint8_t* d[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int32_t c = c0 + 64 * t + lane;
            d[t] = nullptr;
            if (c < total) {
                int i = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    if ((int32_t)S.pre[i + step - 1] <= c) {
                        i += step;
                    }
                }
                int32_t q = c - (i > 0 ? (int32_t)S.pre[i - 1] : 0);
                const int32_t first = (int32_t)S.c0[i];
                const int which = q >= first ? 1 : 0;
                q -= which ? first : 0;
                const int32_t len = (int32_t)S.n[which][i];
                int32_t p = HEAD + 16 * q;
                p = p + 16 > len ? len - 16 : p;
                v[t] = ld16_once<NT>((const uint8_t*)(uintptr_t)S.src[which][i] + p);
                d[t] = (uint8_t*)(uintptr_t)S.dst[which][i] + p;
            }
        }
        if (!headsStored) {
            if constexpr (TWO) {
                head_store(h0, dst0, m0, end1);
            }
            head_store(h1, dst1, m1, end1);
            headsStored = true;
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            if (d[t] != nullptr) {
                st16(d[t], v[t]);
            }
        }
    }
    if (!headsStored) {  // (only big copies this step)
        if constexpr (TWO) {
            head_store(h0, dst0, m0, end1);
        }
        head_store(h1, dst1, m1, end1);
    }
    wave_mem_order();
}

}  // namespace sp

}  // namespace achip
