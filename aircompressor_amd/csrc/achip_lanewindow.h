// achip_lanewindow.h -- a lane's output behind an LDS window, for the lane-per-block decoders lz4_decompress_v6.hip and
// snappy_decompress_v4.hip (design notes there): bytes are appended to a ring column in LDS and leave for the output buffer in aligned
// 64-byte pieces; recent bytes are read back from the ring.  No cross-lane operation.
#pragma once
#include "achip_lanecopy.h"

namespace achip {
namespace sp {

// 16 bytes at p when they lie inside the buffer ending at `end`, else what does (the rest zero; cold)
__device__ __forceinline__ u32x4 safe_ld16(const uint8_t* p, const uint8_t* end)
{
    if (p + 16 <= end) {
        return ld16(p);
    }
    uint32_t w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16 && p + i < end; i++) {
        w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    }
    return u32x4{w[0], w[1], w[2], w[3]};
}

// the lane's output: an LDS ring column of OUT_DW dwords (dword d at ring[(d & (OUT_DW-1)) * 64]) in front of the output buffer.
// Positions are virtual (position + (address & 63)) so that the 64-byte pieces are aligned in memory.
template <int OUT_DW>
struct LaneOutput {
    static constexpr int OUT_BYTES = OUT_DW * 4;
    static constexpr int REACH = OUT_BYTES - 32;  // farthest back-reference read from the ring (a read takes five dwords)
    static_assert((OUT_DW & (OUT_DW - 1)) == 0 && OUT_DW >= 64, "ring size");
    uint32_t* ring;
    uint8_t* outAligned;
    int32_t outBase;
    int32_t opV;       // virtual output position
    int32_t flushedV;  // output flushed up to here (multiple of 64)
    uint32_t carry;    // content of the dword holding opV (valid below opV)

    __device__ __forceinline__ void init(uint32_t* lds, uint8_t* out)
    {
        ring = lds;
        outBase = (int32_t)((uintptr_t)out & 63);
        outAligned = out - outBase;
        opV = outBase;
        flushedV = 0;
        carry = 0;
    }
    __device__ __forceinline__ u32x4 read16(int32_t sV) const
    {
        const int32_t d = sV >> 2;
        const uint32_t r0 = ring[((d + 0) & (OUT_DW - 1)) * 64], r1 = ring[((d + 1) & (OUT_DW - 1)) * 64], r2 = ring[((d + 2) & (OUT_DW - 1)) * 64],
                       r3 = ring[((d + 3) & (OUT_DW - 1)) * 64], r4 = ring[((d + 4) & (OUT_DW - 1)) * 64];
        const uint32_t s = (uint32_t)(sV & 3);
        return u32x4{alignbyte_u32(r1, r0, s), alignbyte_u32(r2, r1, s), alignbyte_u32(r3, r2, s), alignbyte_u32(r4, r3, s)};
    }
    // append c (1..16) bytes, the low bytes of w
    __device__ __forceinline__ void append(u32x4 w, int32_t c)
    {
        const uint32_t sh = (uint32_t)(opV & 3);  // bytes of the current dword already produced
        const uint32_t keep = sh == 0 ? 0u : ((1u << (8 * sh)) - 1u);
        const uint32_t rs = (4u - sh) & 3u;
        const uint32_t d0 = (carry & keep) | (w.x << (8 * sh));
        const uint32_t d1 = sh ? alignbyte_u32(w.y, w.x, rs) : w.y;
        const uint32_t d2 = sh ? alignbyte_u32(w.z, w.y, rs) : w.z;
        const uint32_t d3 = sh ? alignbyte_u32(w.w, w.z, rs) : w.w;
        const uint32_t d4 = sh ? (w.w >> (8 * rs)) : 0u;
        const int32_t d = opV >> 2;
        const int32_t total = (int32_t)sh + c;  // bytes of the stream that are real
        ring[((d + 0) & (OUT_DW - 1)) * 64] = d0;
        if (total > 4) ring[((d + 1) & (OUT_DW - 1)) * 64] = d1;
        if (total > 8) ring[((d + 2) & (OUT_DW - 1)) * 64] = d2;
        if (total > 12) ring[((d + 3) & (OUT_DW - 1)) * 64] = d3;
        if (total > 16) ring[((d + 4) & (OUT_DW - 1)) * 64] = d4;
        const int32_t last = total >> 2;  // dword that holds the new position
        carry = last == 0 ? d0 : (last == 1 ? d1 : (last == 2 ? d2 : (last == 3 ? d3 : d4)));
        opV += c;
        wave_mem_order();
    }
    __device__ __forceinline__ void flush_piece()
    {
        const int32_t d = flushedV >> 2;
        if (flushedV >= outBase) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32x4 g = {ring[((d + 4 * k + 0) & (OUT_DW - 1)) * 64], ring[((d + 4 * k + 1) & (OUT_DW - 1)) * 64], ring[((d + 4 * k + 2) & (OUT_DW - 1)) * 64],
                                 ring[((d + 4 * k + 3) & (OUT_DW - 1)) * 64]};
                *(u32x4*)(outAligned + flushedV + 16 * k) = g;
            }
        }
        else {  // the piece straddling the start of the output buffer (cold)
            for (int32_t p = outBase; p < flushedV + 64; p++) {
                outAligned[p] = (uint8_t)(ring[((p >> 2) & (OUT_DW - 1)) * 64] >> (8 * (p & 3)));
            }
        }
        flushedV += 64;
        wave_mem_order();
    }
    __device__ __forceinline__ void flush_complete()
    {
        while (opV - flushedV >= 64) {
            flush_piece();
        }
    }
    // end of block: what is left of the last piece
    __device__ __forceinline__ void flush_tail()
    {
        flush_complete();
        const int32_t lo = flushedV > outBase ? flushedV : outBase;
        for (int32_t p = lo; p < opV; p++) {
            outAligned[p] = (uint8_t)(ring[((p >> 2) & (OUT_DW - 1)) * 64] >> (8 * (p & 3)));
        }
        wave_mem_order();
    }
};

}  // namespace sp

}  // namespace achip
