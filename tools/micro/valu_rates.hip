// VALU issue cost of the integer instructions the byte / bit kernels are made of (gfx950): per-SIMD clocks per wave64 instruction, measured with 8 wavefronts per SIMD
// and 8 independent chains per lane (throughput, not latency).  hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rates tools/micro/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, int iters, uint32_t seed)
{
    uint64_t a[CHAINS];
    uint32_t s = seed + threadIdx.x;
    for (int c = 0; c < CHAINS; c++) a[c] = (uint64_t)(s * (c + 3)) | ((uint64_t)(s ^ (c * 77u)) << 32);
    const uint32_t sh = (seed & 7) + 1;  // a shift the compiler cannot fold (uniform, in an SGPR)
    const uint32_t vsh = (threadIdx.x & 7) + 1;  // ... and one in a VGPR
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            uint32_t lo = (uint32_t)a[c], hi = (uint32_t)(a[c] >> 32);
            if (OP == 0) { lo = lo + hi; a[c] = ((uint64_t)hi << 32) | lo; }                                  // v_add_u32
            else if (OP == 1) { a[c] = a[c] << vsh; a[c] |= 1; }                                               // v_lshlrev_b64 (+ v_or)
            else if (OP == 2) { a[c] = a[c] >> vsh; a[c] |= 0x8000000000000001ull; }                            // v_lshrrev_b64 (+ 2 v_or)
            else if (OP == 3) { lo = lo * hi + 1u; a[c] = ((uint64_t)hi << 32) | lo; }                          // v_mul_lo_u32 (+ add / mad)
            else if (OP == 4) { a[c] = (uint64_t)lo * (uint64_t)hi + a[c]; }                                    // v_mad_u64_u32
            else if (OP == 5) { lo = __builtin_amdgcn_alignbyte(hi, lo, vsh); a[c] = ((uint64_t)hi << 32) | lo; }  // v_alignbyte_b32
            else if (OP == 6) { lo = __builtin_amdgcn_perm(hi, lo, 0x06050403u + vsh); a[c] = ((uint64_t)hi << 32) | lo; }  // v_perm_b32
            else if (OP == 7) { lo = __builtin_amdgcn_ubfe(lo, vsh, 9u) + hi; a[c] = ((uint64_t)hi << 32) | lo; }  // v_bfe_u32 + add
            else if (OP == 8) { lo = (lo << vsh) | (hi >> (32 - vsh)); a[c] = ((uint64_t)hi << 32) | lo; }      // two 32-bit shifts + or (what a 64-bit shift's upper half is)
            else if (OP == 9) { a[c] = a[c] + ((uint64_t)vsh << 33) + 1; }                                      // 64-bit add (v_add_co + v_addc)
            else if (OP == 10) { lo = (uint32_t)__builtin_clz(lo | 1u) + hi; a[c] = ((uint64_t)hi << 32) | lo; }  // v_ffbh + add
            else if (OP == 11) { a[c] = a[c] << sh; a[c] |= 1; }                                               // v_lshlrev_b64 by an SGPR amount
            else if (OP == 12) { lo = __builtin_amdgcn_alignbit(hi, lo, vsh); a[c] = ((uint64_t)hi << 32) | lo; }  // v_alignbit_b32 (a funnel shift: 64 -> 32 bits)
        }
    }
    uint64_t r = 0;
    for (int c = 0; c < CHAINS; c++) r ^= a[c];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
float run(uint64_t* d, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    uint64_t* d;
    hipMalloc(&d, 2048 * 256 * 8);
    const int iters = 2000;
    const char* names[13] = {"v_add_u32", "v_lshlrev_b64 (VGPR amount) + or", "v_lshrrev_b64 + 2 or", "v_mul_lo_u32 + add", "v_mad_u64_u32", "v_alignbyte_b32", "v_perm_b32", "v_bfe_u32 + add",
                             "32-bit shl + shr + or + sub", "64-bit add (2 instr) x2", "v_ffbh + or + add", "v_lshlrev_b64 (SGPR amount) + or", "v_alignbit_b32"};
    float ms[13] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters), run<6>(d, iters), run<7>(d, iters), run<8>(d, iters), run<9>(d, iters),
                    run<10>(d, iters), run<11>(d, iters), run<12>(d, iters)};
    // 2048 workgroups x 4 wavefronts over 1024 SIMDs = 8 wavefronts per SIMD (one round); a SIMD executes 8 x iters x CHAINS body executions
    for (int i = 0; i < 13; i++) printf("%-40s %8.3f ms  %6.2f clk per body per SIMD (2.4 GHz)\n", names[i], ms[i], ms[i] * 2.4e6 / (8.0 * iters * CHAINS));
    return 0;
}
