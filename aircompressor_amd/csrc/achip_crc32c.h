// achip_crc32c.h -- CRC-32C (Castagnoli) of a buffer by one wavefront, for the x-snappy-framed chunk checksums
// (M/snappy/Crc32C.java:29-50: the Java class is the slicing-by-8 table form of the same function).
//
// A CRC register is linear over GF(2), so the 64 lanes can each take every 64th 32-bit word of the buffer (a coalesced
// 256-byte row per step) and keep a partial register that is advanced by one ROW per step -- four table lookups, the
// same work as slicing-by-4 -- instead of by one word:  acc = ADV256(acc) ^ word.  At the end lane i's register is
// advanced by the 64 - i words that follow its last word, the 64 registers are XORed, and the < 256 bytes behind the
// last full row are added by the byte-wise loop.  "Advance a register by s zero bytes" is  A_0[b0] ^ A_1[b1] ^ A_2[b2] ^
// A_3[b3]  with A_k[b] = advance(b << 8k); the set for 4 bytes is the usual slicing-by-4 table set, the set for 256 bytes
// comes from it by six doublings (apply a set to its own entries).  Both sets live in LDS (8 KB + 4 KB of build space).
#pragma once
#include "achip_device.h"

namespace achip {

struct Crc32cTables {
    uint32_t a4[4][256];    // advance by 4 bytes: a4[3] is the byte-wise table T0
    uint32_t a256[4][256];  // advance by 256 bytes
    uint32_t build[4][256];
};

__device__ __forceinline__ uint32_t crc32c_advance(const uint32_t (*a)[256], uint32_t v)
{
    return a[0][v & 0xFF] ^ a[1][(v >> 8) & 0xFF] ^ a[2][(v >> 16) & 0xFF] ^ a[3][v >> 24];
}

// once per wavefront (the block is one wavefront: __syncthreads is its barrier)
__device__ inline void crc32c_tables_init(Crc32cTables& t, int lane)
{
    for (int i = lane; i < 256; i += 64) {  // T0: one byte into an empty register
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; k++) {
            c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        }
        t.a4[3][i] = c;
    }
    __syncthreads();
    for (int k = 2; k >= 0; k--) {  // a4[k][b] = a4[k + 1][b] advanced by one more byte
        for (int i = lane; i < 256; i += 64) {
            const uint32_t v = t.a4[k + 1][i];
            t.a4[k][i] = (v >> 8) ^ t.a4[3][v & 0xFF];
        }
        __syncthreads();
    }
    // doublings 4 -> 8 -> ... -> 256 bytes, ping-pong between a256 and build (six steps end in a256)
    const uint32_t(*src)[256] = t.a4;
    for (int step = 0; step < 6; step++) {
        uint32_t(*dst)[256] = (step & 1) ? t.a256 : t.build;
        for (int i = lane; i < 1024; i += 64) {
            dst[i >> 8][i & 255] = crc32c_advance(src, src[i >> 8][i & 255]);
        }
        __syncthreads();
        src = dst;
    }
}

// CRC-32C of p[0, n): every lane returns the value
__device__ inline uint32_t wave_crc32c(const Crc32cTables& t, const uint8_t* p, int32_t n, int lane)
{
    const int32_t rows = n >> 8;
    uint32_t reg = 0xFFFFFFFFu;
    if (rows > 0) {
        uint32_t acc = 0;
        for (int32_t r = 0; r < rows; r++) {
            uint32_t w = ld4(p + ((size_t)r << 8) + 4 * lane);
            if (r == 0 && lane == 0) {
                w ^= 0xFFFFFFFFu;  // the initial register meets the first word
            }
            acc = crc32c_advance(t.a256, acc) ^ w;
        }
        for (int k = 0; k < 64 - lane; k++) {  // the words of the last row behind this lane's, and the word itself
            acc = crc32c_advance(t.a4, acc);
        }
        // XOR over the wavefront
        acc ^= __shfl_xor((int)acc, 1);
        acc ^= __shfl_xor((int)acc, 2);
        acc ^= __shfl_xor((int)acc, 4);
        acc ^= __shfl_xor((int)acc, 8);
        acc ^= __shfl_xor((int)acc, 16);
        acc ^= __shfl_xor((int)acc, 32);
        reg = acc;
    }
    for (int32_t i = rows << 8; i < n; i++) {  // (uniform) the bytes behind the last full row
        reg = t.a4[3][(reg ^ p[i]) & 0xFF] ^ (reg >> 8);
    }
    return ~reg;
}

__device__ __forceinline__ uint32_t crc32c_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }  // Crc32C.java:47-50

}  // namespace achip
