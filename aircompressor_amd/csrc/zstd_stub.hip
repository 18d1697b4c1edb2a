// zstd_stub.hip -- placeholder launchers until the Zstd kernels land: every block reports "unsupported".
#include "achip_device.h"

namespace achip {

__global__ void fill_unsupported_kernel(BatchArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.nBlocks) {
        a.outLen[i] = 0;
        a.status[i] = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED);
        a.errOffset[i] = 0;
    }
}

int64_t zstd_compress_scratch_bytes(int32_t) { return 0; }

hipError_t launch_zstd_compress(const BatchArgs& a, hipStream_t stream, void*, int64_t, int)
{
    hipLaunchKernelGGL(fill_unsupported_kernel, dim3((unsigned)((a.nBlocks + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace achip
