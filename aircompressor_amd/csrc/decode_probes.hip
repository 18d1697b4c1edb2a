// decode_probes.hip -- the device-side probes behind the decoders' auto mode: which decoder suits a batch is decided from the
// batch itself, without a host round trip -- every candidate is launched and the ones not chosen return at once (lz4_pick, achip_device.h).
//
//   lz4_mixed_groups_kernel       counts the groups of 16 consecutive blocks (= one ring-decoder wavefront) whose compressed sizes differ by
//                                 more than 2x: long-copy blocks next to text make the ring decoder's lane groups take different paths
//   lz4_sequence_sample_kernel    1024 sampled blocks, the LZ4 sequences at each one's head: bytes per sequence (text: 9 .. 40; long-copy data: >= 100)
//   snappy_element_sample_kernel  the same for Snappy elements
//
// (Until round 4 these lived in lz4_decompress_v5.hip / snappy_decompress_v4.hip beside the lane-per-block decoders, which the two-pass
// decoders superseded: 300 .. 330 GiB/s against 515 on the corpus batch; removed.)
#include "achip_device.h"

namespace achip {

// counts the groups of 16 consecutive blocks whose compressed sizes differ by more than 2x
__global__ __launch_bounds__(256) void lz4_mixed_groups_kernel(BatchArgs a, int32_t* mixedGroups, int32_t minBlocks)
{
    const int32_t n = batch_count(a);
    if (n < minBlocks || n <= 0) {
        return;  // too few blocks for the lane-per-block decoder: the count stays 0
    }
    const int64_t group = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t first = group * 16;
    bool mixed = false;
    if (first < n) {
        int32_t lo = 0x7FFFFFFF, hi = 0;
        for (int64_t b = first; b < first + 16 && b < n; b++) {
            const int32_t len = a.srcLen[b];
            lo = len < lo ? len : lo;
            hi = len > hi ? len : hi;
        }
        mixed = (int64_t)hi > 2 * (int64_t)lo;
    }
    const int found = __popcll(__ballot(mixed));
    if ((threadIdx.x & 63) == 0 && found > 0) {
        atomicAdd(mixedGroups, found);
    }
}

// auto mode, LZ4 only: how long are the sequences?  1024 sampled blocks, the sequences in the first SAMPLE_HEAD bytes of each (at most 96; a
// lane per sample; headers only).  The head of the block is staged in LDS first: parsing it straight from global memory was one dependent
// round trip per header byte -- 0.11 .. 0.14 ms in front of every decode call, 1.5 % of the headline's step.
constexpr int SAMPLE_HEAD = 768;
constexpr int SAMPLE_STRIDE = SAMPLE_HEAD + 4;  // (an odd number of dwords: the lanes' heads start in different banks)

// the first min(inLimit, SAMPLE_HEAD) bytes of a lane's block into its LDS row (16-byte loads that stay inside the stream)
__device__ __forceinline__ int32_t sample_stage_head(uint8_t* h, const uint8_t* __restrict__ in, int32_t inLimit)
{
    const int32_t limit = inLimit < SAMPLE_HEAD ? inLimit : SAMPLE_HEAD;
#pragma unroll 4
    for (int32_t p = 0; p < limit; p += 16) {
        if (p + 16 <= inLimit) {
            const u32x4 v = ld16(in + p);
            __builtin_memcpy(h + p, &v, 16);
        }
        else {
            for (int32_t i = p; i < limit; i++) {
                h[i] = in[i];
            }
        }
    }
    return limit;
}

__global__ __launch_bounds__(64) void lz4_sequence_sample_kernel(BatchArgs a, int32_t* stats, int32_t minBlocks, int32_t shortLimit)
{
    const int32_t n = batch_count(a);
    if (n < minBlocks || n <= 0) {
        return;
    }
    __shared__ __attribute__((aligned(16))) uint8_t heads[64 * SAMPLE_STRIDE];
    const int32_t t = blockIdx.x * 64 + threadIdx.x;
    const int64_t block = (int64_t)t * n / 1024;
    uint8_t* const h = heads + threadIdx.x * SAMPLE_STRIDE;
    const int32_t inLimit = sample_stage_head(h, a.srcBase + a.srcOff[block], a.srcLen[block]);
    int64_t ip = 0;  // 64-bit: a run of length-extension bytes must not wrap the cursor
    int32_t seqs = 0;
    int64_t bytes = 0;
    while (ip < inLimit && seqs < 96) {
        const int32_t token = h[ip++];
        int64_t lit = token >> 4;
        if (lit == 15) {
            int32_t v = 255;
            while (v == 255 && ip < inLimit) {
                v = h[ip++];
                lit += v;
            }
        }
        bytes += lit;
        ip += lit;
        seqs++;
        if (ip + 2 > inLimit) {
            break;  // last literals, the end of the head (or nonsense: the decoders will say)
        }
        ip += 2;
        int64_t ml = token & 15;
        if (ml == 15) {
            int32_t v = 255;
            while (v == 255 && ip < inLimit) {
                v = h[ip++];
                ml += v;
            }
        }
        bytes += ml + 4;
    }
    bytes = bytes < 0 || bytes > (1 << 24) ? (1 << 24) : bytes;
    atomicAdd(stats + 1, seqs);
    atomicAdd(stats + 2, (int32_t)(bytes >> 2));  // (in units of 4 bytes: 1024 samples x 16 MiB stay inside 32 bits)
    if (shortLimit > 0) {  // the verdict block by block (lz4_pick, achip_device.h): [4] sampled blocks of short sequences, [5] sampled blocks
        const unsigned long long isShort = __ballot(seqs > 0 && (bytes >> 2) < (int64_t)shortLimit * seqs);
        if (threadIdx.x == 0) {
            atomicAdd(stats + 4, (int32_t)__popcll(isShort));
            atomicAdd(stats + 5, 64);
        }
    }
}

// shortLimit > 0 (the batched block API): also count the sampled BLOCKS whose own sequences are short (bytes / 4 per sequence below the limit)
hipError_t launch_lz4_sequence_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks, int32_t shortLimit)
{
    hipLaunchKernelGGL(lz4_sequence_sample_kernel, dim3(16), dim3(64), 0, stream, a, stats, minBlocks, shortLimit);
    return hipGetLastError();
}

hipError_t launch_lz4_mixed_groups(const BatchArgs& a, hipStream_t stream, int32_t* mixedGroups, int32_t minBlocks)
{
    const hipError_t e = hipMemsetAsync(mixedGroups, 0, 8 * sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(((a.nBlocks + 15) / 16 + 255) / 256);
    hipLaunchKernelGGL(lz4_mixed_groups_kernel, dim3(grid), dim3(256), 0, stream, a, mixedGroups, minBlocks);
    return hipGetLastError();
}

__device__ __forceinline__ int32_t snappy_op_entry_probe(int32_t op)  // opLookupTable layout :223-271
{
    const int32_t kind = op & 3;
    const int32_t hi = op >> 2;
    if (kind == 0) {
        return hi < 60 ? hi + 1 : (((hi - 59) << 11) | 1);
    }
    if (kind == 1) {
        return (1 << 11) | ((hi >> 3) << 8) | ((hi & 7) + 4);
    }
    return ((kind == 2 ? 2 : 4) << 11) | (hi + 1);
}

// auto mode: how long are the elements?  1024 sampled blocks, the elements in the first 768 bytes of each (at most 192; a lane per sample;
// tags only), parsed from an LDS copy of the block's head (the reason: lz4_sequence_sample_kernel above)
__global__ __launch_bounds__(64) void snappy_element_sample_kernel(BatchArgs a, int32_t* stats, int32_t minBlocks, int32_t shortLimit)
{
    const int32_t n = batch_count(a);
    if (n < minBlocks || n <= 0) {
        return;
    }
    constexpr int HEAD = 768, STRIDE = HEAD + 4;
    __shared__ __attribute__((aligned(16))) uint8_t heads[64 * STRIDE];
    const int32_t t = blockIdx.x * 64 + threadIdx.x;
    const int64_t block = (int64_t)t * n / 1024;
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    const int32_t inLen = a.srcLen[block];
    uint8_t* const h = heads + threadIdx.x * STRIDE;
    const int32_t inLimit = inLen < HEAD ? inLen : HEAD;
#pragma unroll 4
    for (int32_t p = 0; p < inLimit; p += 16) {
        if (p + 16 <= inLen) {
            const u32x4 v = ld16(in + p);
            __builtin_memcpy(h + p, &v, 16);
        }
        else {
            for (int32_t i = p; i < inLimit; i++) {
                h[i] = in[i];
            }
        }
    }
    int32_t ip = 0, elements = 0;
    int64_t bytes = 0;
    while (ip < inLimit && ip < 5 && (h[ip] & 0x80) != 0) {  // the length preamble
        ip++;
    }
    ip++;
    while (ip < inLimit && elements < 192) {
        const int32_t opc = h[ip++];
        const int32_t entry = snappy_op_entry_probe(opc);
        const int32_t trailerBytes = entry >> 11;
        if (ip + trailerBytes > inLimit) {
            break;
        }
        uint32_t trailer = 0;
        for (int i = 0; i < trailerBytes; i++) {
            trailer |= (uint32_t)h[ip + i] << (8 * i);
        }
        ip += trailerBytes;
        int64_t length = entry & 0xff;
        if ((opc & 3) == 0) {
            length += trailer;
            ip += (int32_t)(length < (int64_t)(inLimit - ip) ? length : inLimit - ip);
        }
        bytes += length;
        elements++;
    }
    bytes = bytes < 0 || bytes > (1 << 24) ? (1 << 24) : bytes;
    atomicAdd(stats + 1, elements);
    atomicAdd(stats + 2, (int32_t)(bytes >> 2));
    if (shortLimit > 0) {
        const unsigned long long isShort = __ballot(elements > 0 && (bytes >> 2) < (int64_t)shortLimit * elements);
        if (threadIdx.x == 0) {
            atomicAdd(stats + 4, (int32_t)__popcll(isShort));
            atomicAdd(stats + 5, 64);
        }
    }
}

hipError_t launch_snappy_element_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks, int32_t shortLimit)
{
    hipLaunchKernelGGL(snappy_element_sample_kernel, dim3(16), dim3(64), 0, stream, a, stats, minBlocks, shortLimit);
    return hipGetLastError();
}

}  // namespace achip
