// achip_lanes.h -- lane-per-block LDS rings for the streaming LZ77 decoders (v3 kernels).
//
// The group-per-block decoders (achip_rings.h) run 16 independent state machines per wavefront and pay for every path
// any of them takes.  Here every LANE owns a block (64 blocks per wavefront) and every block advances by the same
// straight-line step -- "parse what is due, then move up to 16 bytes" -- so the wavefront stays converged and an
// instruction does useful work for 64 blocks instead of 16 x 4 lanes of one.
//
// LDS layout: lane-interleaved columns.  Dword d of lane L's ring lives at word (d * 64 + L): a wavefront access
// touches 64 different banks whatever the lanes' ring positions are -- conflict-free by construction.
//   * input ring  (IN_DW dwords per lane):  the compressed stream, pulled from HBM in 16-byte aligned granules, one
//     granule requested ahead (registers) so the HBM latency overlaps a step;
//   * output ring (OUT_DW dwords per lane): the history window.  Bytes are produced with whole-dword stores: the tail
//     dword's unused high bytes are garbage that the next copy overwrites, the head dword is merged with `carry` (the
//     lane's copy of the dword under the write position), so there is no read-modify-write and no byte store.
//     Complete 16-byte granules are flushed with one 16-byte store per lane; L2 assembles the 128-byte lines.
// Positions are virtual (position + (address & 15)) so that granules are 16-byte aligned in memory.
// No cross-lane operation is used anywhere: lanes may leave at any time, and the code runs unchanged on a CPU
// one lane at a time (tools/hostemu), which is how the test suite checks it on a CPU before it goes to the GPU.
#pragma once
#include "achip_device.h"

namespace achip {

template <int IN_DW, int OUT_DW>
struct LaneRings {
    static constexpr int IN_BYTES = IN_DW * 4, OUT_BYTES = OUT_DW * 4;
    static constexpr int REACH = OUT_BYTES - 24;  // farthest back-reference served from LDS (the copy in flight may scribble 19 bytes ahead)
    static_assert((IN_DW & (IN_DW - 1)) == 0 && (OUT_DW & (OUT_DW - 1)) == 0 && IN_DW >= 16 && OUT_DW >= 16, "ring sizes");

    uint32_t* inR;   // this lane's column of the input ring: dword d at inR[(d & (IN_DW-1)) * 64]
    uint32_t* outR;
    const uint8_t* inAligned;
    uint8_t* outAligned;
    int32_t inBase, outBase;
    int32_t inEndV;     // virtual end of the input
    int32_t inLoadedV;  // the ring holds virtual [inLoadedV - IN_BYTES, inLoadedV)
    int32_t opV;        // virtual output position
    int32_t flushedV;   // output flushed up to here (multiple of 16)
    uint32_t carry;     // content of the dword holding opV (valid below opV)
    u32x4 pending;      // the granule at inLoadedV, requested one refill ahead

    __device__ __forceinline__ void init(uint32_t* ldsIn, uint32_t* ldsOut, const uint8_t* in, int32_t inLimit, uint8_t* out)
    {
        inR = ldsIn;
        outR = ldsOut;
        inBase = (int32_t)((uintptr_t)in & 15);
        outBase = (int32_t)((uintptr_t)out & 15);
        inAligned = in - inBase;
        outAligned = out - outBase;
        inEndV = inLimit + inBase;
        inLoadedV = 0;
        opV = outBase;
        flushedV = 0;
        carry = 0;
        pending = fetch_granule(0);
    }
    __device__ __forceinline__ int32_t op() const { return opV - outBase; }

    // 16-byte granule at virtual position v of the input; bytes outside the input read as 0
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
    // make input bytes [.., pos + need) resident (need <= IN_BYTES - 16); bytes past the end read as 0
    __device__ __forceinline__ void ensure_input(int32_t pos, int32_t need)
    {
        const int32_t want = pos + inBase + need;
        while (want > inLoadedV && inLoadedV < inEndV) {
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
    // 8 input bytes at position pos (resident)
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

    // 16 bytes at virtual byte position sV of a ring column
    template <int DW>
    static __device__ __forceinline__ u32x4 ring_read16(const uint32_t* ring, int32_t sV)
    {
        const int32_t d = sV >> 2;
        const uint32_t r0 = ring[((d + 0) & (DW - 1)) * 64], r1 = ring[((d + 1) & (DW - 1)) * 64], r2 = ring[((d + 2) & (DW - 1)) * 64],
                       r3 = ring[((d + 3) & (DW - 1)) * 64], r4 = ring[((d + 4) & (DW - 1)) * 64];
        const uint32_t s = (uint32_t)(sV & 3);
        return u32x4{alignbyte_u32(r1, r0, s), alignbyte_u32(r2, r1, s), alignbyte_u32(r3, r2, s), alignbyte_u32(r4, r3, s)};
    }

    // append c (1..16) bytes, the low bytes of w, at the output position
    __device__ __forceinline__ void append(u32x4 w, int32_t c)
    {
        const uint32_t sh = (uint32_t)(opV & 3);  // bytes of the current dword already produced
        const uint32_t keep = (1u << (8 * sh)) - 1u;
        // stream = carry's low sh bytes followed by w: dword k = (w[k] : w[k-1]) >> 8*(4-sh)
        const uint32_t rs = (4u - sh) & 3u;
        uint32_t d0 = (carry & keep) | (w.x << (8 * sh));
        uint32_t d1 = sh ? alignbyte_u32(w.y, w.x, rs) : w.y;
        uint32_t d2 = sh ? alignbyte_u32(w.z, w.y, rs) : w.z;
        uint32_t d3 = sh ? alignbyte_u32(w.w, w.z, rs) : w.w;
        uint32_t d4 = sh ? (w.w >> (8 * rs)) : 0u;
        const int32_t d = opV >> 2;
        const int32_t total = (int32_t)sh + c;  // bytes of the stream that are real
        outR[((d + 0) & (OUT_DW - 1)) * 64] = d0;
        if (total > 4) outR[((d + 1) & (OUT_DW - 1)) * 64] = d1;
        if (total > 8) outR[((d + 2) & (OUT_DW - 1)) * 64] = d2;
        if (total > 12) outR[((d + 3) & (OUT_DW - 1)) * 64] = d3;
        if (total > 16) outR[((d + 4) & (OUT_DW - 1)) * 64] = d4;
        const int32_t last = total >> 2;  // dword that holds the new position
        carry = last == 0 ? d0 : (last == 1 ? d1 : (last == 2 ? d2 : (last == 3 ? d3 : d4)));
        opV += c;
        wave_mem_order();
        if (opV - flushedV >= 16) {
            flush_granule();
        }
    }
    __device__ __forceinline__ void flush_granule()
    {
        const int32_t d = flushedV >> 2;
        const u32x4 g = {outR[((d + 0) & (OUT_DW - 1)) * 64], outR[((d + 1) & (OUT_DW - 1)) * 64], outR[((d + 2) & (OUT_DW - 1)) * 64],
                         outR[((d + 3) & (OUT_DW - 1)) * 64]};
        if (flushedV >= outBase) {
            *(u32x4*)(outAligned + flushedV) = g;
        }
        else {  // the granule straddling the start of the output buffer (cold)
            const uint32_t w[4] = {g.x, g.y, g.z, g.w};
            for (int32_t p = outBase; p < flushedV + 16; p++) {
                outAligned[p] = (uint8_t)(w[(p - flushedV) >> 2] >> (8 * (p & 3)));
            }
        }
        flushedV += 16;
        wave_mem_order();
    }
    // end of block: the bytes of the last, partial granule
    __device__ __forceinline__ void flush_tail()
    {
        while (opV - flushedV >= 16) {
            flush_granule();
        }
        const int32_t d = flushedV >> 2;
        const uint32_t w[4] = {outR[((d + 0) & (OUT_DW - 1)) * 64], outR[((d + 1) & (OUT_DW - 1)) * 64], outR[((d + 2) & (OUT_DW - 1)) * 64],
                               outR[((d + 3) & (OUT_DW - 1)) * 64]};
        const int32_t lo = flushedV > outBase ? flushedV : outBase;
        for (int32_t p = lo; p < opV; p++) {
            outAligned[p] = (uint8_t)(w[(p - flushedV) >> 2] >> (8 * (p & 3)));
        }
        wave_mem_order();
    }

    // one step (<= 16 bytes) of a literal run: input bytes at pos (resident) -> output
    __device__ __forceinline__ void copy_literals_step(int32_t pos, int32_t c) { append(ring_read16<IN_DW>(inR, pos + inBase), c); }

    // one step of a back-reference of `dist` bytes (dist >= c: the caller splits overlapping copies)
    __device__ __forceinline__ void copy_match_step(int32_t dist, int32_t c)
    {
        u32x4 w;
        if (dist <= REACH) {
            w = ring_read16<OUT_DW>(outR, opV - dist);
        }
        else {
            w = ld16(outAligned + (opV - dist));  // flushed at least 16 bytes ago: OUT_BYTES >= 64
        }
        append(w, c);
    }
};

}  // namespace achip
