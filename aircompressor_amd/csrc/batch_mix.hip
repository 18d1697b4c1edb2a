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

}  // namespace achip
