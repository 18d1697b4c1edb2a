/*
 * oracle/misc.c -- synthetic data generator + batch drivers.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   orc_random_generator follows T/snappy/RandomGenerator.java:25-74 driven by
 *   java.util.Random(301) (LCG: seed = (seed * 0x5DEECE66D + 0xB) mod 2^48,
 *   next(bits) = seed >>> (48 - bits), nextInt(256) = (256 * next(31)) >> 31).
 *   T/ = src/test/java/io/airlift/compress/v3/
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <string.h>

static uint64_t jseed;
static void jrandom_init(int64_t seed) { jseed = ((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1); }
static int32_t jrandom_next(int bits)
{
    jseed = (jseed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
    return (int32_t)(int64_t)(jseed >> (48 - bits));
}
static int32_t jrandom_next_int_256(void) { return (int32_t)((256LL * (int64_t)jrandom_next(31)) >> 31); }

void orc_random_generator(double compression_ratio, uint8_t* out, int64_t len)
{
    /* data = new byte[1048576 + 100]; fragments of 100 bytes until 1048576; positions past that stay 0 */
    uint8_t raw[100];
    jrandom_init(301);
    memset(out, 0, (size_t)len);
    for (int64_t i = 0; i < 1048576 && i < len; i += 100) {
        int32_t n = (int32_t)(100 * compression_ratio);
        if (n < 1) n = 1;
        for (int32_t k = 0; k < n; k++) raw[k] = (uint8_t)jrandom_next_int_256();
        for (int32_t j = 0; j < 100 && i + j < len;) {
            int32_t chunk = n < 100 - j ? n : 100 - j;
            for (int32_t k = 0; k < chunk && i + j + k < len; k++) out[i + j + k] = raw[k];
            j += chunk;
        }
    }
}

int64_t orc_batch(int32_t op, const uint8_t* src_base, const int64_t* src_off, const int32_t* src_len,
                  uint8_t* dst_base, const int64_t* dst_off, const int32_t* dst_cap,
                  int32_t* out_len, int32_t* status, int64_t* err_off, int32_t n_blocks)
{
    int64_t total = 0;
    for (int32_t i = 0; i < n_blocks; i++) {
        const uint8_t* s = src_base + src_off[i];
        uint8_t* d = dst_base + dst_off[i];
        int64_t eo = 0;
        int64_t r;
        switch (op) {
            case ACHIP_OP_LZ4_DECOMPRESS: r = orc_lz4_decompress(s, src_len[i], d, dst_cap[i], &eo); break;
            case ACHIP_OP_LZ4_COMPRESS: r = orc_lz4_compress(s, src_len[i], d, dst_cap[i]); break;
            case ACHIP_OP_SNAPPY_DECOMPRESS: r = orc_snappy_decompress(s, src_len[i], d, dst_cap[i], &eo); break;
            case ACHIP_OP_SNAPPY_COMPRESS: r = orc_snappy_compress(s, src_len[i], d, dst_cap[i]); break;
            case ACHIP_OP_ZSTD_DECOMPRESS: r = orc_zstd_decompress(s, src_len[i], d, dst_cap[i], &eo); break;
            case ACHIP_OP_ZSTD_COMPRESS: r = orc_zstd_compress(s, src_len[i], d, dst_cap[i]); break;
            default: r = ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT); break;
        }
        if (r >= 0) {
            if (out_len) out_len[i] = (int32_t)r;
            if (status) status[i] = 0;
            if (err_off) err_off[i] = 0;
            total += r;
        }
        else {
            if (out_len) out_len[i] = 0;
            if (status) status[i] = (int32_t)r;
            if (err_off) err_off[i] = eo;
        }
    }
    return total;
}
