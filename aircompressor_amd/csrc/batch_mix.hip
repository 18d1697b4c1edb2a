// batch_mix.hip -- the codec-bucketing scheduler of mixed batches ("mixed batches are first bucketed by codec so each launch
// is homogeneous"; BASELINE configs[4]).  The host sorts the item indices by codec op (stable); `gather` builds the per-op descriptor
// arrays from the caller's arrays in that order, every op's kernels then run over their contiguous slice, and `scatter` puts the
// results back where the caller's item order wants them.  Blocks stay where they are: only 44 bytes of metadata per item move.
#include "achip_device.h"

namespace achip {

__global__ __launch_bounds__(256) void mix_gather_kernel(const int32_t* __restrict__ perm, int32_t n, const int64_t* __restrict__ srcOff,
                                                         const int32_t* __restrict__ srcLen, const int64_t* __restrict__ dstOff,
                                                         const int32_t* __restrict__ dstCap, int64_t* __restrict__ gSrcOff,
                                                         int32_t* __restrict__ gSrcLen, int64_t* __restrict__ gDstOff, int32_t* __restrict__ gDstCap)
{
    const int32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) {
        const int32_t i = perm[j];
        gSrcOff[j] = srcOff[i];
        gSrcLen[j] = srcLen[i];
        gDstOff[j] = dstOff[i];
        gDstCap[j] = dstCap[i];
    }
}

__global__ __launch_bounds__(256) void mix_scatter_kernel(const int32_t* __restrict__ perm, int32_t n, const int32_t* __restrict__ gOutLen,
                                                          const int32_t* __restrict__ gStatus, const int64_t* __restrict__ gErr,
                                                          int32_t* __restrict__ outLen, int32_t* __restrict__ status, int64_t* __restrict__ errOffset)
{
    const int32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) {
        const int32_t i = perm[j];
        outLen[i] = gOutLen[j];
        status[i] = gStatus[j];
        if (errOffset != nullptr) {
            errOffset[i] = gErr[j];
        }
    }
}

hipError_t launch_mix_gather(const int32_t* perm, int32_t n, const BatchArgs& a, int64_t* gSrcOff, int32_t* gSrcLen, int64_t* gDstOff, int32_t* gDstCap, hipStream_t stream)
{
    hipLaunchKernelGGL(mix_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, perm, n, a.srcOff, a.srcLen, a.dstOff, a.dstCap, gSrcOff, gSrcLen, gDstOff, gDstCap);
    return hipGetLastError();
}

hipError_t launch_mix_scatter(const int32_t* perm, int32_t n, const int32_t* gOutLen, const int32_t* gStatus, const int64_t* gErr, const BatchArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(mix_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, perm, n, gOutLen, gStatus, gErr, a.outLen, a.status, a.errOffset);
    return hipGetLastError();
}

// A copy by a kernel (the host-pointer pipeline, achip_abi.cpp: option host.blit): dst / src may be pinned HOST memory -- hipHostMalloc'ed slots are
// mapped into the device's address space --, 16 bytes per lane, a KiB per wavefront and instruction.  Why not hipMemcpyAsync: the pipeline's uploads and
// downloads, on two streams, did not overlap on the link (the times of the two directions ADDED: profiles/r06_notes.md); a copy engine one way and a kernel
// the other do.  bytes need not be a multiple of 16; dst and src are 16-byte aligned (slot offsets are multiples of 256).
__global__ __launch_bounds__(256) void blit_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int64_t bytes)
{
    const int64_t whole = bytes >> 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < whole; i += stride) {
        st16(dst + 16 * i, ld16(src + 16 * i));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(bytes & 15)) {
        dst[16 * whole + threadIdx.x] = src[16 * whole + threadIdx.x];
    }
}

hipError_t launch_blit(void* dst, const void* src, int64_t bytes, int workgroups, hipStream_t stream)
{
    if (bytes <= 0) {
        return hipSuccess;
    }
    const int64_t need = ((bytes >> 4) + 255) / 256;
    const unsigned grid = (unsigned)(need < 1 ? 1 : (need < workgroups ? need : workgroups));
    hipLaunchKernelGGL(blit_kernel, dim3(grid), dim3(256), 0, stream, (uint8_t*)dst, (const uint8_t*)src, bytes);
    return hipGetLastError();
}

}  // namespace achip
