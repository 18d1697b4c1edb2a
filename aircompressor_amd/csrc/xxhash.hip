// xxhash.hip -- batched XXH64 / XXH32 for gfx950 (the public xxhash package on the GPU).
//
// Replaces XxHash64Hasher.hash(MemorySegment, long) (M/xxhash/XxHash64Hasher.java:78-86 -> XxHash64JavaHasher.java:126)
// and XxHash32Hasher.hash(MemorySegment, int) (M/xxhash/XxHash32Hasher.java -> XxHash32JavaHasher.java:112) for many
// buffers per call -- the checksums of the containers built on the block codecs (Zstd frame checksum, LZ4 frame header /
// block / content checksums).  Both functions keep four accumulators over 32- / 16-byte stripes; an accumulator is one
// serial multiply-rotate chain, so a buffer is hashed by FOUR lanes (one per accumulator) and a wavefront hashes 16
// buffers: the 4 lanes of a buffer read one stripe (32 or 16 contiguous bytes) per step, four steps in flight.
// Roofline: HBM (read-once); algorithmic bytes = the buffer lengths.
#include "achip_xxhash.h"

namespace achip {

struct HashArgs {
    const uint8_t* __restrict__ srcBase;
    const int64_t* __restrict__ srcOff;
    const int32_t* __restrict__ srcLen;
    int32_t n;
};

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
