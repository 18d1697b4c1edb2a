// lz4_decompress_v5.hip -- batched LZ4 block decode for gfx950: lane-per-block PARSE, sequence-parallel EXECUTE.
//
// Same contract and the same Java-order checks as lz4_decompress_v2.hip (M/lz4/Lz4RawDecompressor.java:35-198).
//
// The ring decoders (v2) spend their time issuing instructions: a wavefront walks 16 blocks, one sequence of each at a
// time, and executes the union of the paths those 16 state machines take (profiles/r01_notes.md: 2.3x the instructions
// of a converged wavefront; ~128 SIMD cycles per sequence on text).  Here the two halves of the work are separated so
// that each is converged:
//   PARSE    every lane owns a block (64 blocks per wavefront) and walks its token stream without touching the literal
//            bytes: token, length extensions, offset, the Java checks in their order.  Up to K sequence records
//            {literal source, literal length, match length, offset} per block go to LDS.  One record per trip for
//            every lane, whatever the lengths are.
//   EXECUTE  the wavefront then takes 64/K blocks at a time, one LANE PER SEQUENCE: a row scan of the lengths gives every
//            sequence its output position; all literal runs are copied at once (they depend on nothing); matches are
//            resolved in rounds (a match may run once its source lies below the block's high-water mark -- the start of
//            the first match still pending -- or when it is that first match); in text most matches reach back farther
//            than the 16 sequences of a row, so a row takes two or three rounds.  Copies are exact (no scribbling: the
//            neighbouring bytes belong to another lane).  Long copies are done by the whole wavefront.
// Output goes straight to HBM (L2 combines the pieces); there is no output ring.
#include "achip_device.h"

namespace achip {
namespace sp {

// the lane's window on its compressed stream: an LDS ring column fed with aligned 16-byte granules, one requested ahead
template <int IN_DW>
struct LaneInput {
    static constexpr int IN_BYTES = IN_DW * 4;
    static_assert((IN_DW & (IN_DW - 1)) == 0 && IN_DW >= 8, "ring size");
    uint32_t* inR;  // dword d of this lane's ring at inR[(d & (IN_DW-1)) * 64]
    const uint8_t* inAligned;
    int32_t inBase;
    int32_t inEndV;
    int32_t inLoadedV;  // virtual [.., inLoadedV) is in the ring (as far back as the ring reaches)
    u32x4 pending;      // the granule at inLoadedV

    __device__ __forceinline__ void init(uint32_t* lds, const uint8_t* in, int32_t inLimit)
    {
        inR = lds;
        inBase = (int32_t)((uintptr_t)in & 15);
        inAligned = in - inBase;
        inEndV = inLimit + inBase;
        inLoadedV = 0;
        pending = fetch_granule(0);
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
    // make [pos, pos + need) resident (need <= IN_BYTES - 16; bytes past the end read as 0).  The stream is only read
    // forwards, so a jump over literal bytes restarts the ring at the granule of pos.
    __device__ __forceinline__ void ensure_input(int32_t pos, int32_t need)
    {
        const int32_t v = pos + inBase;
        if (v >= inLoadedV + 16) {
            inLoadedV = v & ~15;
            pending = fetch_granule(inLoadedV);
        }
        while (v + need > inLoadedV && inLoadedV < inEndV) {
            const int32_t d = inLoadedV >> 2;
            inR[((d + 0) & (IN_DW - 1)) * 64] = pending.x;
            inR[((d + 1) & (IN_DW - 1)) * 64] = pending.y;
            inR[((d + 2) & (IN_DW - 1)) * 64] = pending.z;
            inR[((d + 3) & (IN_DW - 1)) * 64] = pending.w;
            inLoadedV += 16;
            pending = fetch_granule(inLoadedV);
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

constexpr int REC_STRIDE = 65;  // dwords between consecutive (record, field) rows: spreads a block's records over the banks
constexpr int HEAD = 32;        // bytes of a copy its own lane moves; the rest of all copies is dealt out to the lanes in 16-byte chunks

struct CopyScratch {  // LDS, per wavefront
    uint32_t pre[64], n[64];
    uint64_t dst[64], src[64];
};

// One wavefront-wide copy step: every active lane has n bytes to move from src to dst; the ranges of a lane do not
// overlap and every source is final.  Exact (no byte outside [dst, dst + n) is written, none outside [src, src + n) read).
// All loads of a step are issued before its stores, so a step costs about one memory round trip however the lengths are
// distributed: <= HEAD bytes by the lane itself (two overlapping 16-byte pieces, or 8/4/2/1), the remainders as 16-byte
// chunks handed out evenly (a binary search over the running chunk count finds a chunk's owner), the last chunk of a copy
// ending exactly at its end.
__device__ __forceinline__ void copy_step(CopyScratch& S, int lane, bool active, uint8_t* dst, const uint8_t* src, int32_t n)
{
    const int32_t m = active ? (n < HEAD ? n : HEAD) : 0;
    const bool wide = m >= 16;
    u32x4 A = {0, 0, 0, 0}, B = {0, 0, 0, 0};
    uint64_t v8 = 0;
    uint32_t v4 = 0, v2 = 0, v1 = 0;
    if (wide) {
        A = ld16(src);
        B = ld16(src + m - 16);
    }
    else {
        if (m & 8) v8 = ld8(src);
        if (m & 4) v4 = ld4(src + (m & 8));
        if (m & 2) v2 = ld2(src + (m & 12));
        if (m & 1) v1 = src[m & 14];
    }
    const int32_t chunks = (active && n > HEAD) ? (n - HEAD + 15) >> 4 : 0;
    const int32_t incl = wave_scan(chunks, lane);
    const int32_t total = __builtin_amdgcn_readlane(incl, 63);
    if (total > 0) {  // (uniform) publish the copies for the chunk loop below
        S.pre[lane] = (uint32_t)incl;
        S.n[lane] = (uint32_t)n;
        S.dst[lane] = (uint64_t)(uintptr_t)dst;
        S.src[lane] = (uint64_t)(uintptr_t)src;
    }
    if (wide) {
        st16(dst, A);
        st16(dst + m - 16, B);
    }
    else {
        if (m & 8) st8(dst, v8);
        if (m & 4) st4(dst + (m & 8), v4);
        if (m & 2) st2(dst + (m & 12), v2);
        if (m & 1) dst[m & 14] = (uint8_t)v1;
    }
    if (total > 0) {
        wave_mem_order();
        for (int32_t c0 = 0; c0 < total; c0 += 256) {
            u32x4 v[4];
            uint8_t* d[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
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
                    const int32_t before = i > 0 ? (int32_t)S.pre[i - 1] : 0;
                    const int32_t len = (int32_t)S.n[i];
                    int32_t p = HEAD + 16 * (c - before);
                    p = p + 16 > len ? len - 16 : p;
                    v[t] = ld16((const uint8_t*)(uintptr_t)S.src[i] + p);
                    d[t] = (uint8_t*)(uintptr_t)S.dst[i] + p;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (d[t] != nullptr) {
                    st16(d[t], v[t]);
                }
            }
        }
        wave_mem_order();
    }
}

}  // namespace sp

// K: records per block per round (8 or 16); 64 / K blocks are executed side by side
template <int IN_DW, int K>
__global__ __launch_bounds__(64) void lz4_decompress_seqpar_kernel(BatchArgs a)
{
    using namespace sp;
    constexpr int ROWS = 64 / K;
    __shared__ uint32_t ldsIn[IN_DW * 64];
    __shared__ uint32_t ldsRec[4 * K * REC_STRIDE];
    __shared__ CopyScratch S;
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    const bool have = block < a.nBlocks;
    const uint8_t* in = have ? a.srcBase + a.srcOff[block] : a.srcBase;
    uint8_t* out = have ? a.dstBase + a.dstOff[block] : a.dstBase;
    const int32_t inLimit = have ? a.srcLen[block] : 0;
    const int32_t outLimit = have ? a.dstCap[block] : 0;

    LaneInput<IN_DW> R;
    R.init(ldsIn + lane, in, inLimit);

    int32_t st = 0;
    int32_t eo = 0;
    int32_t ip = 0;
    int32_t op = 0;
    bool done = !have;
    const int32_t fastOutLimit = outLimit - 8;

#define LZ4_FAIL(detail, off)                          \
    {                                                  \
        st = mk_status(ACHIP_CLASS_MALFORMED, detail); \
        eo = (int32_t)(off);                           \
        done = true;                                   \
    }

    if (have) {
        if (inLimit == 0) {  // :48-50
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_LZ4_INPUT_EMPTY);
            done = true;
        }
        else if (outLimit == 0) {  // :52-57 (the Java method returns -1 here)
            if (!(inLimit == 1 && in[0] == 0)) {
                st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT);
            }
            done = true;
        }
    }

    while (__ballot(!done) != 0) {
        // ---------------- PARSE: up to K records per lane ----------------
        const int32_t opStart = op;
        int32_t cnt = 0;
        for (int k = 0; k < K; k++) {
            if (__ballot(!done) == 0) {
                break;
            }
            if (!done) {
                if (ip >= inLimit) {  // the Java loop condition :59
                    done = true;
                }
                else {
                    R.ensure_input(ip, 12);
                    uint64_t w = R.in_u64(ip);
                    const int32_t token = (int32_t)(w & 0xFF);
                    ip++;
                    int32_t lit = token >> 4;  // :62-77
                    if (lit == 0xF) {
                        if (ip >= inLimit) {
                            LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                        }
                        else {
                            int32_t v;
                            do {
                                R.ensure_input(ip, 4);
                                v = (int32_t)R.in_u8(ip++);
                                lit = (int32_t)((uint32_t)lit + (uint32_t)v);
                            } while (v == 255 && ip < inLimit - 15);
                        }
                    }
                    if (!done && lit < 0) {
                        LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                    }
                    bool lastLiterals = false;
                    if (!done) {
                        const int64_t litEnd = (int64_t)ip + lit;
                        const int64_t litOutLimit = (int64_t)op + lit;
                        if (litOutLimit > fastOutLimit - 4 || litEnd > inLimit - 8) {  // :82-96 last literals
                            if (litOutLimit > outLimit) {
                                LZ4_FAIL(ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE, ip);
                            }
                            else if (litEnd != inLimit) {
                                LZ4_FAIL(ACHIP_D_LZ4_INPUT_NOT_CONSUMED, ip);
                            }
                            else {
                                lastLiterals = true;
                            }
                        }
                    }
                    if (!done) {
                        const int32_t litSrc = ip;
                        ip += lit;
                        op += lit;
                        int32_t ml = 0;
                        int32_t offset = 0;
                        if (lastLiterals) {
                            done = true;
                        }
                        else {
                            R.ensure_input(ip, 12);
                            w = R.in_u64(ip);
                            offset = (int32_t)(w & 0xFFFF);  // :113-119
                            ip += 2;
                            if (offset == 0 || offset > op) {
                                LZ4_FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, ip);
                            }
                            else {
                                ml = token & 0xF;  // :122-138
                                bool bad = false;
                                if (ml == 0xF) {
                                    int32_t v;
                                    do {
                                        if (ip > inLimit - 5) {
                                            bad = true;
                                            break;
                                        }
                                        R.ensure_input(ip, 4);
                                        v = (int32_t)R.in_u8(ip++);
                                        ml = (int32_t)((uint32_t)ml + (uint32_t)v);
                                    } while (v == 255);
                                }
                                ml = (int32_t)((uint32_t)ml + 4u);
                                if (bad || ml < 0) {
                                    LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                                }
                                else {
                                    const int64_t matchOutLimit = (int64_t)op + ml;
                                    if (matchOutLimit > fastOutLimit - 4 && matchOutLimit > outLimit - 5) {  // :168-171
                                        LZ4_FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, ip);
                                    }
                                }
                            }
                            if (done) {  // the sequence failed after its literals: Java has copied them, nobody can tell
                                ml = 0;
                            }
                            op += ml;
                        }
                        uint32_t* r = ldsRec + (cnt * 4) * REC_STRIDE + lane;
                        r[0] = (uint32_t)litSrc;
                        r[REC_STRIDE] = (uint32_t)lit;
                        r[2 * REC_STRIDE] = (uint32_t)ml;
                        r[3 * REC_STRIDE] = (uint32_t)offset;
                        cnt++;
                    }
                }
            }
        }
        wave_mem_order();

        // ---------------- EXECUTE: ROWS blocks at a time, one lane per sequence ----------------
        const unsigned long long busy = a.ringPad == 240 ? 0ull : __ballot(cnt > 0);  // (ring pad 240 / 224: timing aids -- parse only / no matches)
        const int j = lane & (K - 1);
        const int segBase = lane & ~(K - 1);
        for (int first = 0; first < 64; first += ROWS) {
            if (((busy >> first) & ((1ull << ROWS) - 1ull)) == 0) {
                continue;
            }
            const int b = first + lane / K;  // the block (= parse lane) this lane works for
            const int32_t bCnt = __shfl(cnt, b);
            const int32_t bOp = __shfl(opStart, b);
            const uint8_t* const bIn = (const uint8_t*)(uintptr_t)shfl_u64((uint64_t)(uintptr_t)in, b);
            uint8_t* const bOut = (uint8_t*)(uintptr_t)shfl_u64((uint64_t)(uintptr_t)out, b);
            const bool valid = j < bCnt;
            const uint32_t* r = ldsRec + (j * 4) * REC_STRIDE + b;
            const int32_t litSrc = valid ? (int32_t)r[0] : 0;
            const int32_t lit = valid ? (int32_t)r[REC_STRIDE] : 0;
            int32_t rem = valid ? (int32_t)r[2 * REC_STRIDE] : 0;
            int32_t dist = valid ? (int32_t)r[3 * REC_STRIDE] : 0;
            const int32_t incl = seg_scan<K>(lit + rem, j);
            uint8_t* const dstLit = bOut + bOp + (incl - lit - rem);
            uint8_t* cur = dstLit + lit;  // where the match (what is left of it) goes

            // literal runs: nothing depends on them
            copy_step(S, lane, lit > 0, dstLit, bIn + litSrc, lit);

            // matches, in rounds.  `cur` of the first pending match of a block is its high-water mark: everything below is
            // final.  A match whose source ends below the mark runs; the first pending match always runs -- if it overlaps
            // itself (dist < rem) one period now, and as the written part repeats the period, twice as much the next round.
            for (;;) {
                const unsigned long long pm = a.ringPad == 224 ? 0ull : __ballot(rem > 0);
                if (pm == 0) {
                    break;
                }
                const uint32_t segMask = (uint32_t)(pm >> segBase) & ((1u << K) - 1u);
                const int firstPending = segBase + (segMask ? __builtin_ctz(segMask) : 0);
                const uint64_t mark = shfl_u64((uint64_t)(uintptr_t)cur, firstPending);
                const bool ready = rem > 0 && (lane == firstPending || (uint64_t)(uintptr_t)(cur - dist + rem) <= mark);
                const int32_t n = ready ? (rem < dist ? rem : dist) : 0;
                copy_step(S, lane, n > 0, cur, cur - dist, n);
                cur += n;
                rem -= n;
                if (n > 0 && rem > 0 && dist < (1 << 28)) {
                    dist += dist;
                }
            }
        }
        wave_mem_order();
    }
#undef LZ4_FAIL
    if (have) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

hipError_t launch_lz4_decompress_seqpar(const BatchArgs& a, hipStream_t stream)
{
    const unsigned grid = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL((lz4_decompress_seqpar_kernel<8, 8>), dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace achip
