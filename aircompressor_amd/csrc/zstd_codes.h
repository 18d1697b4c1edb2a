/* zstd_codes.h -- sequence code -> (baseline, extra bits) as arithmetic instead of table lookups, so the
 * lane-per-item FSE loop of zstd_decompress_pipe.hip has no dependent memory access after the state lookup.
 * Equal to the tables of M/zstd/ZstdFrameDecompressor.java:68-83 and M/zstd/Constants.java:66-78 for every
 * code (tests/test_host_logic.py compiles this header with gcc and compares all 36 + 53 + 29 entries).
 * Plain C so that the check needs no GPU toolchain. */
#ifndef ACHIP_ZSTD_CODES_H
#define ACHIP_ZSTD_CODES_H
#include <stdint.h>
#ifdef __HIPCC__
#define ACHIP_CODES_FN __device__ __forceinline__
#else
#define ACHIP_CODES_FN static inline
#endif

ACHIP_CODES_FN void achip_zstd_ll_code(int32_t code, int32_t* base, int32_t* bits)
{
    if (code < 16) {
        *base = code;
        *bits = 0;
    }
    else if (code < 25) {
        const int i = code - 16;
        *bits = (int32_t)((0x433221111ull >> (4 * i)) & 0xF);
        *base = i < 8 ? (int32_t)((0x28201C1816141210ull >> (8 * i)) & 0xFF) : 48;
    }
    else {
        *bits = code - 19;
        *base = 1 << *bits;
    }
}

ACHIP_CODES_FN void achip_zstd_ml_code(int32_t code, int32_t* base, int32_t* bits)
{
    if (code < 32) {
        *base = code + 3;
        *bits = 0;
    }
    else if (code < 43) {
        const int i = code - 32;
        *bits = (int32_t)((0x54433221111ull >> (4 * i)) & 0xF);
        *base = i < 8 ? (int32_t)((0x3B332F2B29272523ull >> (8 * i)) & 0xFF) : (int32_t)((0x635343u >> (8 * (i - 8))) & 0xFF);
    }
    else {
        *bits = code - 36;
        *base = (1 << *bits) + 3;
    }
}

ACHIP_CODES_FN int32_t achip_zstd_of_base(int32_t code) { return code < 2 ? code : (1 << code) - 3; }

#endif
