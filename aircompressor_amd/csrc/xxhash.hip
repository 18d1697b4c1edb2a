// xxhash.hip -- batched XXH64 / XXH32 for gfx950 (the public xxhash package on the GPU).
//
// Replaces XxHash64Hasher.hash(MemorySegment, long) (M/xxhash/XxHash64Hasher.java:78-86 -> XxHash64JavaHasher.java:126)
// and XxHash32Hasher.hash(MemorySegment, int) (M/xxhash/XxHash32Hasher.java -> XxHash32JavaHasher.java:112) for many
// buffers per call -- the checksums of the containers built on the block codecs (Zstd frame checksum, LZ4 frame header /
// block / content checksums).  Both functions keep four accumulators over 32- / 16-byte stripes; an accumulator is one
// serial multiply-rotate chain, so a buffer is hashed by FOUR lanes (one per accumulator) and a wavefront hashes 16
// buffers: the 4 lanes of a buffer read one stripe (32 or 16 contiguous bytes) per step, four steps in flight.
// Roofline: HBM (read-once); algorithmic bytes = the buffer lengths.
#include "achip_device.h"

namespace achip {

struct HashArgs {
    const uint8_t* __restrict__ srcBase;
    const int64_t* __restrict__ srcOff;
    const int32_t* __restrict__ srcLen;
    int32_t n;
};

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

__global__ __launch_bounds__(64) void xxh64_batch_kernel(HashArgs a, uint64_t seed, int64_t* __restrict__ out)
{
    const int lane = threadIdx.x;
    const int s = lane & 3;
    const int64_t i = (int64_t)blockIdx.x * 16 + (lane >> 2);
    if (i >= a.n) {
        return;  // whole quads leave together
    }
    const int32_t len = a.srcLen[i];
    const uint64_t h = quad_xxh64(a.srcBase + a.srcOff[i], len < 0 ? 0 : len, seed, s, lane - s);
    if (s == 0) {
        out[i] = (int64_t)h;
    }
}

__global__ __launch_bounds__(64) void xxh32_batch_kernel(HashArgs a, uint32_t seed, int32_t* __restrict__ out)
{
    const int lane = threadIdx.x;
    const int s = lane & 3;
    const int64_t i = (int64_t)blockIdx.x * 16 + (lane >> 2);
    if (i >= a.n) {
        return;
    }
    const int32_t len = a.srcLen[i];
    const uint32_t h = quad_xxh32(a.srcBase + a.srcOff[i], len < 0 ? 0 : len, seed, s, lane - s);
    if (s == 0) {
        out[i] = (int32_t)h;
    }
}

hipError_t launch_xxh64_batch(const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, int32_t n, uint64_t seed, int64_t* out, hipStream_t stream)
{
    if (n <= 0) {
        return hipSuccess;
    }
    HashArgs a{(const uint8_t*)srcBase, srcOff, srcLen, n};
    hipLaunchKernelGGL(xxh64_batch_kernel, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, stream, a, seed, out);
    return hipGetLastError();
}

hipError_t launch_xxh32_batch(const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, int32_t n, uint32_t seed, int32_t* out, hipStream_t stream)
{
    if (n <= 0) {
        return hipSuccess;
    }
    HashArgs a{(const uint8_t*)srcBase, srcOff, srcLen, n};
    hipLaunchKernelGGL(xxh32_batch_kernel, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, stream, a, seed, out);
    return hipGetLastError();
}

}  // namespace achip
