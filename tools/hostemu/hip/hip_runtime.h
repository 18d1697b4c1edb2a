// Host shim used ONLY by tools/hostemu: lets the single-lane (GS=1) instantiations of the decoder
// kernels be compiled with g++ and stepped through on the CPU while debugging.  Not part of the product.
#pragma once
#include <cstdint>
#include <cstring>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
extern thread_local dim3 threadIdx, blockIdx, blockDim;
typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)              \
    do {                                                                         \
        blockDim = block;                                                        \
        for (unsigned bx_ = 0; bx_ < dim3(grid).x; bx_++)                        \
            for (unsigned tx_ = 0; tx_ < dim3(block).x; tx_++) {                 \
                blockIdx = dim3(bx_);                                            \
                threadIdx = dim3(tx_);                                           \
                kernel(__VA_ARGS__);                                             \
            }                                                                    \
    } while (0)
inline void __syncthreads() {}
inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
// cross-lane builtins that only the NOT emulated kernels of a shared header use (they must compile, they never run here)
inline int __builtin_amdgcn_update_dpp(int old, int src, int, int, int, bool) { (void)old; return src; }
inline int __builtin_amdgcn_readlane(int v, int) { return v; }
template <typename T> inline T atomicAdd(T* p, T v) { T old = *p; *p += v; return old; }
template <typename T> inline T __shfl(T v, int) { return v; }

// dynamic shared memory of the emulated launch: one arena, re-used by every "workgroup"
static uint8_t hostemu_dynamic_lds[1 << 20] __attribute__((aligned(64)));
inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
