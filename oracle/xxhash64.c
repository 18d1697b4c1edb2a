/*
 * oracle/xxhash64.c -- XXH64 as the reference's Zstd path computes it.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Follows M/zstd/XxHash64.java:182-291 (hash, updateBody, updateTail, finalShuffle).
 */
#include "oracle.h"
#include <string.h>

#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL

static inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t mix(uint64_t cur, uint64_t v) { return rotl(cur + v * P2, 31) * P1; }          /* :246-249 */
static inline uint64_t upd(uint64_t h, uint64_t v) { return (h ^ mix(0, v)) * P1 + P4; }              /* :251-255 */

uint64_t orc_xxh64(const uint8_t* in, int64_t len, uint64_t seed)
{
    uint64_t hash;
    const uint8_t* p = in;
    if (len >= 32) { /* updateBody :220-244 */
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        int64_t remaining = len;
        while (remaining >= 32) {
            uint64_t a, b, c, d;
            memcpy(&a, p, 8);
            memcpy(&b, p + 8, 8);
            memcpy(&c, p + 16, 8);
            memcpy(&d, p + 24, 8);
            v1 = mix(v1, a);
            v2 = mix(v2, b);
            v3 = mix(v3, c);
            v4 = mix(v4, d);
            p += 32;
            remaining -= 32;
        }
        hash = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        hash = upd(hash, v1);
        hash = upd(hash, v2);
        hash = upd(hash, v3);
        hash = upd(hash, v4);
    }
    else {
        hash = seed + P5;
    }
    hash += (uint64_t)len;

    int64_t index = len & ~(int64_t)31; /* updateTail :199-218 */
    while (index <= len - 8) {
        uint64_t v;
        memcpy(&v, in + index, 8);
        hash = rotl(hash ^ mix(0, v), 27) * P1 + P4;
        index += 8;
    }
    if (index <= len - 4) {
        uint32_t v;
        memcpy(&v, in + index, 4);
        hash = rotl(hash ^ ((uint64_t)v * P1), 23) * P2 + P3;
        index += 4;
    }
    while (index < len) {
        hash = rotl(hash ^ ((uint64_t)in[index] * P5), 11) * P1;
        index++;
    }
    hash ^= hash >> 33; /* finalShuffle :281-289 */
    hash *= P2;
    hash ^= hash >> 29;
    hash *= P3;
    hash ^= hash >> 32;
    return hash;
}
