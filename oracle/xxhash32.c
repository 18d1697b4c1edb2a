/*
 * oracle/xxhash32.c -- XXH32 as the reference's public xxhash package computes it (also the checksum of
 * the LZ4 frame container, M/lz4/Lz4FrameCompression.java:106,236).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Follows M/xxhash/XxHash32JavaHasher.java:68-110 (hash), :343-366 (mix, updateTail x2, finalShuffle).
 * Pinned by T/xxhash/TestXxHash32.java:45-46 ("" -> 0x02CC5D05, "abc" -> 0x32D153FF) in tests/test_oracle_golden.py.
 */
#include "oracle.h"
#include <string.h>

#define P1 0x9E3779B1u
#define P2 0x85EBCA77u
#define P3 0xC2B2AE3Du
#define P4 0x27D4EB2Fu
#define P5 0x165667B1u

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t mix32(uint32_t cur, uint32_t v) { return rotl32(cur + v * P2, 13) * P1; } /* :343-346 */

uint32_t orc_xxh32(const uint8_t* in, int64_t len, uint32_t seed)
{
    const uint8_t* p = in;
    const uint8_t* end = in + len;
    uint32_t hash;
    if (len >= 16) { /* :76-91 */
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        while (p <= end - 16) {
            uint32_t a, b, c, d;
            memcpy(&a, p, 4);
            memcpy(&b, p + 4, 4);
            memcpy(&c, p + 8, 4);
            memcpy(&d, p + 12, 4);
            v1 = mix32(v1, a);
            v2 = mix32(v2, b);
            v3 = mix32(v3, c);
            v4 = mix32(v4, d);
            p += 16;
        }
        hash = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    }
    else {
        hash = seed + P5; /* :92-94 */
    }
    hash += (uint32_t)len; /* :96 */
    while (p + 4 <= end) { /* :99-102, updateTail(int) :348-351 */
        uint32_t w;
        memcpy(&w, p, 4);
        hash = rotl32(hash + w * P3, 17) * P4;
        p += 4;
    }
    while (p < end) { /* :104-107, updateTail(byte) :353-357 */
        hash = rotl32(hash + (uint32_t)(*p) * P5, 11) * P1;
        p++;
    }
    hash ^= hash >> 15; /* finalShuffle :359-366 */
    hash *= P2;
    hash ^= hash >> 13;
    hash *= P3;
    hash ^= hash >> 16;
    return hash;
}
