// LDS throughput of 16-byte accesses at byte-unaligned addresses (gfx950), against aligned ones and against four aligned dword accesses:
// what a 16-bytes-per-lane copy through the decoders' LDS rings would cost.  hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_unaligned tools/micro/lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const int* mis, u32x4* out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf[37888];
    const int t = threadIdx.x;
    for (int i = t * 16; i < 37888; i += 256 * 16) *(u32x4*)(buf + i) = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    // a group of four lanes owns 592 bytes (as the ring kernels' groups do); lane g of a group touches 16 bytes at 16 g + m
    const int grp = t >> 2, g = t & 3;
    unsigned char* base = buf + grp * 592;
    int m = mis[t];  // per GROUP misalignment (same for the four lanes), 0..15
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        const int pos = (it * 64 + m) & 255;  // walks a 256-byte ring
        if (MODE == 0) {  // aligned 16-byte read + aligned 16-byte write
            u32x4 v = *(const u32x4*)(base + ((pos & ~15) + 16 * g) % 256);
            v.x += it;
            *(u32x4*)(base + 288 + ((pos & ~15) + 16 * g) % 256) = v;
            acc.x ^= v.y;
        }
        else if (MODE == 1) {  // unaligned 16-byte read + aligned write
            u32x4 v;
            __builtin_memcpy(&v, base + (pos + 16 * g) % 256, 16);
            v.x += it;
            *(u32x4*)(base + 288 + ((pos & ~15) + 16 * g) % 256) = v;
            acc.x ^= v.y;
        }
        else if (MODE == 2) {  // unaligned read + unaligned write
            u32x4 v;
            __builtin_memcpy(&v, base + (pos + 16 * g) % 256, 16);
            v.x += it;
            __builtin_memcpy(base + 288 + (pos + 5 + 16 * g) % 256, &v, 16);
            acc.x ^= v.y;
        }
        else if (MODE == 3) {  // today's move: per lane four (2 aligned dword reads + alignbyte) + four aligned dword writes, dwords interleaved over the group
            unsigned w[4];
            for (int q = 0; q < 4; q++) {
                const int pv = pos + 4 * (g + 4 * q);
                const int a = pv & ~3;
                const unsigned lo = *(const unsigned*)(base + (a & 255));
                const unsigned hi = *(const unsigned*)(base + ((a + 4) & 255));
                w[q] = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(pv & 3));
            }
            w[0] += it;
            for (int q = 0; q < 4; q++) *(unsigned*)(base + 288 + (((pos & ~3) + 4 * (g + 4 * q)) & 255)) = w[q];
            acc.x ^= w[1];
        }
        else if (MODE == 4) {  // unaligned write only
            u32x4 v = {(unsigned)it, acc.x, 2u, 3u};
            __builtin_memcpy(base + 288 + (pos + 16 * g) % 256, &v, 16);
        }
        else if (MODE == 5) {  // unaligned 8-byte writes x 2
            unsigned long long a = it, b = acc.x;
            __builtin_memcpy(base + 288 + (pos + 16 * g) % 256, &a, 8);
            __builtin_memcpy(base + 288 + (pos + 8 + 16 * g) % 256, &b, 8);
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + t] = acc + *(u32x4*)(buf + 16 * t);
}

template <int MODE>
float run(const int* dMis, u32x4* dOut, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, dMis, dOut, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, dMis, dOut, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int iters = 20000;
    int* dMis;
    u32x4* dOut;
    hipMalloc(&dMis, 256 * 4);
    hipMalloc(&dOut, 1024 * 256 * 16);
    const char* names[6] = {"aligned 16 B read + aligned 16 B write", "UNALIGNED 16 B read + aligned write", "UNALIGNED read + UNALIGNED write", "today: 4 x (2 dword reads + alignbyte) + 4 dword writes",
                            "UNALIGNED 16 B write only", "2 x UNALIGNED 8 B write only"};
    for (int pattern = 0; pattern < 6; pattern++) {  // misalignment per group: all 0 / all 5 / random bytes / all 4 / all 8 / random multiples of 4
        std::vector<int> mis(256);
        for (int t = 0; t < 256; t++) mis[t] = pattern == 0 ? 0 : (pattern == 1 ? 5 : (pattern == 2 ? ((t >> 2) * 7 + 3) & 15 : (pattern == 3 ? 4 : (pattern == 4 ? 8 : (((t >> 2) * 7 + 3) & 3) * 4))));
        hipMemcpy(dMis, mis.data(), 256 * 4, hipMemcpyHostToDevice);
        const char* pn[6] = {"0 (aligned positions)", "5 for every group", "differs per group (bytes)", "4 for every group", "8 for every group", "differs per group (multiples of 4)"};
        printf("misalignment %s\n", pn[pattern]);
        float ms[6] = {run<0>(dMis, dOut, iters), run<1>(dMis, dOut, iters), run<2>(dMis, dOut, iters), run<3>(dMis, dOut, iters), run<4>(dMis, dOut, iters), run<5>(dMis, dOut, iters)};
        // 1024 workgroups x 4 wavefronts on 256 CUs = 16 wavefronts per CU; cycles per wavefront-iteration at 2.4 GHz, per CU: ms * 2.4e6 / (iters * 16)
        for (int mIdx = 0; mIdx < 6; mIdx++) printf("  %-62s %8.2f ms  %6.1f clk per wavefront-trip (per CU)\n", names[mIdx], ms[mIdx], ms[mIdx] * 2.4e6 / (iters * 16.0));
    }
    return 0;
}
