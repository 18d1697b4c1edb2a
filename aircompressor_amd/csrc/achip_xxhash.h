// achip_xxhash.h -- XXH64 / XXH32 of one buffer by a quad of lanes (one accumulator per lane); used by xxhash.hip
// (the batched hashers) and zstd_decompress_pipe.hip (frame checksums).
#pragma once
#include "achip_device.h"

namespace achip {

// XXH64 of [p, p + len): the calling lane is accumulator s (0..3) of its buffer; `base` is the first lane of the buffer's
// quad.  All four lanes return the hash.  (The same routine checks Zstd frame checksums in zstd_decompress_pipe.hip.)
__device__ __forceinline__ uint64_t quad_xxh64(const uint8_t* __restrict__ p, int32_t len, uint64_t seed, int s, int base)
{
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto mix = [&](uint64_t cur, uint64_t v) { return rotl(cur + v * P2, 31) * P1; };
    uint64_t hash;
    if (len >= 32) {  // XxHash64JavaHasher.java:84-104
        uint64_t v = seed + (s == 0 ? P1 + P2 : (s == 1 ? P2 : (s == 2 ? 0 : (0 - P1))));
        const int32_t stripes = len >> 5;
        const uint8_t* q = p + s * 8;
        int32_t k = 0;
        for (; k + 4 <= stripes; k += 4) {
            const uint64_t x0 = ld8(q + (int64_t)k * 32), x1 = ld8(q + (int64_t)k * 32 + 32), x2 = ld8(q + (int64_t)k * 32 + 64), x3 = ld8(q + (int64_t)k * 32 + 96);
            v = mix(v, x0);
            v = mix(v, x1);
            v = mix(v, x2);
            v = mix(v, x3);
        }
        for (; k < stripes; k++) {
            v = mix(v, ld8(q + (int64_t)k * 32));
        }
        const uint64_t v1 = __shfl(v, base), v2 = __shfl(v, base + 1), v3 = __shfl(v, base + 2), v4 = __shfl(v, base + 3);
        hash = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        hash = (hash ^ mix(0, v1)) * P1 + P4;
        hash = (hash ^ mix(0, v2)) * P1 + P4;
        hash = (hash ^ mix(0, v3)) * P1 + P4;
        hash = (hash ^ mix(0, v4)) * P1 + P4;
    }
    else {
        hash = seed + P5;
    }
    hash += (uint64_t)len;
    int32_t index = len & ~31;  // updateTail :106-124
    while (index <= len - 8) {
        hash = rotl(hash ^ mix(0, ld8(p + index)), 27) * P1 + P4;
        index += 8;
    }
    if (index <= len - 4) {
        hash = rotl(hash ^ ((uint64_t)ld4(p + index) * P1), 23) * P2 + P3;
        index += 4;
    }
    while (index < len) {
        hash = rotl(hash ^ ((uint64_t)p[index] * P5), 11) * P1;
        index++;
    }
    hash ^= hash >> 33;  // finalShuffle
    hash *= P2;
    hash ^= hash >> 29;
    hash *= P3;
    hash ^= hash >> 32;
    return hash;
}

// XXH32: XxHash32JavaHasher.java:68-110, :343-366
__device__ __forceinline__ uint32_t quad_xxh32(const uint8_t* __restrict__ p, int32_t len, uint32_t seed, int s, int base)
{
    constexpr uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, P4 = 0x27D4EB2Fu, P5 = 0x165667B1u;
    auto rotl = [](uint32_t x, int r) { return (x << r) | (x >> (32 - r)); };
    auto mix = [&](uint32_t cur, uint32_t v) { return rotl(cur + v * P2, 13) * P1; };
    uint32_t hash;
    if (len >= 16) {
        uint32_t v = seed + (s == 0 ? P1 + P2 : (s == 1 ? P2 : (s == 2 ? 0u : (0u - P1))));
        const int32_t stripes = len >> 4;
        const uint8_t* q = p + s * 4;
        int32_t k = 0;
        for (; k + 4 <= stripes; k += 4) {
            const uint32_t x0 = ld4(q + (int64_t)k * 16), x1 = ld4(q + (int64_t)k * 16 + 16), x2 = ld4(q + (int64_t)k * 16 + 32), x3 = ld4(q + (int64_t)k * 16 + 48);
            v = mix(v, x0);
            v = mix(v, x1);
            v = mix(v, x2);
            v = mix(v, x3);
        }
        for (; k < stripes; k++) {
            v = mix(v, ld4(q + (int64_t)k * 16));
        }
        const uint32_t v1 = (uint32_t)__shfl((int)v, base), v2 = (uint32_t)__shfl((int)v, base + 1), v3 = (uint32_t)__shfl((int)v, base + 2), v4 = (uint32_t)__shfl((int)v, base + 3);
        hash = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    }
    else {
        hash = seed + P5;
    }
    hash += (uint32_t)len;
    int32_t index = len & ~15;
    while (index <= len - 4) {
        hash = rotl(hash + ld4(p + index) * P3, 17) * P4;
        index += 4;
    }
    while (index < len) {
        hash = rotl(hash + (uint32_t)p[index] * P5, 11) * P1;
        index++;
    }
    hash ^= hash >> 15;
    hash *= P2;
    hash ^= hash >> 13;
    hash *= P3;
    hash ^= hash >> 16;
    return hash;
}

// ---- XXH64 over data that arrives in pieces (XxHash64.java:60-180: update / hash): the four accumulators, the length so far and the pending
// stripe live in memory between the pieces.  update() is a wavefront's call (wave-uniform; lanes 0..3 own an accumulator each, the others
// wait); digest() may be called by any lane after the wavefront's memory is in order. ----
struct Xxh64Stream {
    uint64_t total;
    uint64_t v[4];
    int32_t tailLen;
    int32_t pad;
    uint8_t tail[32];
};
__device__ __forceinline__ void xxh64_stream_reset(Xxh64Stream* s, int lane)
{
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL;
    if (lane == 0) {
        s->total = 0;
        s->v[0] = P1 + P2;
        s->v[1] = P2;
        s->v[2] = 0;
        s->v[3] = 0 - P1;
        s->tailLen = 0;
        s->pad = 0;
    }
    wave_sync();
}
__device__ __forceinline__ void xxh64_stream_update(Xxh64Stream* s, const uint8_t* __restrict__ data, int32_t n, int lane)
{
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto mix = [&](uint64_t cur, uint64_t v) { return rotl(cur + v * P2, 31) * P1; };
    wave_sync();
    const bool works = lane < 4;
    int32_t tailLen = s->tailLen;
    uint64_t v = works ? s->v[lane] : 0;
    int32_t at = 0;
    if (tailLen > 0) {  // (uniform) complete the pending stripe first
        const int32_t take = 32 - tailLen < n ? 32 - tailLen : n;
        if (works) {
            for (int32_t i = lane; i < take; i += 4) {
                s->tail[tailLen + i] = data[i];
            }
        }
        wave_sync();
        tailLen += take;
        at = take;
        if (tailLen == 32) {
            if (works) {
                v = mix(v, ld8(s->tail + 8 * lane));
            }
            tailLen = 0;
        }
    }
    const int32_t stripes = (n - at) >> 5;
    if (works) {
        const uint8_t* q = data + at + 8 * lane;
        for (int32_t k = 0; k < stripes; k++) {
            v = mix(v, ld8(q + (int64_t)k * 32));
        }
    }
    at += stripes * 32;
    wave_sync();
    if (at < n) {  // (uniform; the pending stripe was consumed or there was none: the rest starts a new one)
        if (works) {
            for (int32_t i = lane; i < n - at; i += 4) {
                s->tail[i] = data[at + i];
            }
        }
        tailLen = n - at;
    }
    if (works) {
        s->v[lane] = v;
    }
    if (lane == 0) {
        s->tailLen = tailLen;
        s->total += (uint64_t)(uint32_t)n;
    }
    wave_sync();
}
__device__ __forceinline__ uint64_t xxh64_stream_digest(const Xxh64Stream* s)
{
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto mix = [&](uint64_t cur, uint64_t v) { return rotl(cur + v * P2, 31) * P1; };
    uint64_t hash;
    if (s->total >= 32) {
        const uint64_t v1 = s->v[0], v2 = s->v[1], v3 = s->v[2], v4 = s->v[3];
        hash = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        hash = (hash ^ mix(0, v1)) * P1 + P4;
        hash = (hash ^ mix(0, v2)) * P1 + P4;
        hash = (hash ^ mix(0, v3)) * P1 + P4;
        hash = (hash ^ mix(0, v4)) * P1 + P4;
    }
    else {
        hash = P5;  // (seed 0)
    }
    hash += s->total;
    const uint8_t* t = s->tail;
    const int32_t tailLen = s->tailLen;
    int32_t index = 0;
    while (index <= tailLen - 8) {
        hash = rotl(hash ^ mix(0, ld8(t + index)), 27) * P1 + P4;
        index += 8;
    }
    if (index <= tailLen - 4) {
        hash = rotl(hash ^ ((uint64_t)ld4(t + index) * P1), 23) * P2 + P3;
        index += 4;
    }
    while (index < tailLen) {
        hash = rotl(hash ^ ((uint64_t)t[index] * P5), 11) * P1;
        index++;
    }
    hash ^= hash >> 33;
    hash *= P2;
    hash ^= hash >> 29;
    hash *= P3;
    hash ^= hash >> 32;
    return hash;
}

}  // namespace achip
