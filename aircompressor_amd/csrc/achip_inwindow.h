// achip_inwindow.h -- a wavefront's LDS window over the input it is encoding (lz4_compress_v3.hip, snappy_compress_v3.hip): the input's
// bytes [lo, hi), at most WIN of them (a power of two), kept as a ring indexed by position, refilled CHUNK = WIN / 2 bytes at a time -- one
// memory round trip per CHUNK of progress instead of one per batch of probes.  Reads of up to 24 bytes never wrap: the ring's first
// MIRROR bytes are kept again behind its end.  Behind the input's end the window holds zeros.
#pragma once
#include "achip_device.h"

namespace achip {
namespace inwin {

constexpr int32_t MIRROR = 32;
template <int WIN> constexpr int32_t bytes() { return WIN + MIRROR; }

// 16 bytes at position p (p .. p + 16 inside the window, or behind hi where nobody looks at them): three aligned 8-byte reads, shifted into place
template <int WIN>
__device__ __forceinline__ void read16(const uint8_t* win, int32_t p, uint64_t& first, uint64_t& second)
{
    const uint32_t i = (uint32_t)p & (uint32_t)(WIN - 1);
    const uint32_t a = i & ~7u;
    const uint64_t w0 = *(const uint64_t*)(win + a);
    const uint64_t w1 = *(const uint64_t*)(win + a + 8);
    const uint64_t w2 = *(const uint64_t*)(win + a + 16);
    const uint32_t sh = (i & 7u) * 8u;
    first = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
    second = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;
}

template <int WIN>
__device__ __forceinline__ uint8_t read1(const uint8_t* win, int32_t p) { return win[(uint32_t)p & (uint32_t)(WIN - 1)]; }

// Make the window hold [.., need) for a batch whose first byte is `first` (need - first <= CHUNK; positions only move forward):
// whole chunks are appended; when the batch lies a chunk or more ahead of the window (batches in between read from memory), the window
// starts again at the batch's chunk.  Wave-uniform arguments; the caller orders the LDS stores before its reads.  Returns whether anything was stored.
template <int WIN>
__device__ __forceinline__ bool cover(uint8_t* win, const uint8_t* __restrict__ in, int32_t limit, int32_t first, int32_t need, int32_t& lo, int32_t& hi, int lane)
{
    constexpr int32_t CHUNK = WIN / 2;  // at most 64 lanes x 16 bytes
    static_assert(CHUNK <= 1024 && CHUNK >= 64 && (WIN & (WIN - 1)) == 0, "window size");
    if (hi >= need) {
        return false;
    }
    if (first >= hi + CHUNK) {
        hi = first & ~(CHUNK - 1);
        lo = hi;
    }
    while (hi < need) {
        const int32_t p = hi + lane * 16;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (lane * 16 >= CHUNK) {
        }
        else if (p + 16 <= limit) {
            v = ld16(in + p);
        }
        else if (p < limit) {  // the input's last bytes: zeros behind them
            uint32_t word[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int32_t j = 0; j < 16; j++) {
                if (p + j < limit) {
                    word[j >> 2] |= (uint32_t)in[p + j] << (8 * (j & 3));
                }
            }
            v = u32x4{word[0], word[1], word[2], word[3]};
        }
        const uint32_t w = (uint32_t)p & (uint32_t)(WIN - 1);
        if (lane * 16 < CHUNK) {
            *(u32x4*)(win + w) = v;
            if (w < (uint32_t)MIRROR) {
                *(u32x4*)(win + WIN + w) = v;
            }
        }
        hi += CHUNK;
        lo = lo > hi - WIN ? lo : hi - WIN;
    }
    return true;
}

// number of equal leading bytes of two 8-byte little-endian words
__device__ __forceinline__ int32_t eq_lead(uint64_t a, uint64_t b)
{
    const uint64_t d = a ^ b;
    return d == 0 ? 8 : (__builtin_ctzll(d) >> 3);
}
// number of equal TRAILING bytes (the bytes right before a position, nearest first)
__device__ __forceinline__ int32_t eq_trail(uint64_t a, uint64_t b)
{
    const uint64_t d = a ^ b;
    return d == 0 ? 8 : (__builtin_clzll(d) >> 3);
}

}  // namespace inwin
}  // namespace achip
