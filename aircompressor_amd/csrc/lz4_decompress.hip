// lz4_decompress.hip -- batched LZ4 block decode for gfx950.
//
// Replaces Lz4RawDecompressor.decompress (M/lz4/Lz4RawDecompressor.java:35-198) for a
// batch of independent blocks.  One GS-lane group of a wavefront per block (see
// achip_device.h).  The token grammar is walked exactly as the Java loop does -- the
// same checks, in the same order, with the same thresholds -- so the per-block status
// and error offset are the ones the Java decoder would throw; the byte moves are the
// Java copies' *semantics* (LZ77 byte-sequential), done 16 bytes per lane.
//
// Memory: compressed stream read from HBM (the serial reads are same-address across the
// group = one transaction; literal runs are read GS*16 bytes per step), output written
// straight to HBM, history (match sources) re-read through L2 -- the write-through L1
// never holds a block's own fresh output.  No LDS: a 64 KiB history per block in LDS
// would cap residency at 2 blocks per CU; DESIGN.md section 4 has the measured trade.
#include "achip_device.h"

namespace achip {

template <int GS>
__global__ __launch_bounds__(256) void lz4_decompress_kernel(BatchArgs a)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    const int g = threadIdx.x & (GS - 1);
    const int64_t block = (int64_t)blockIdx.x * GROUPS_PER_WG + (threadIdx.x / GS);
    if (block >= a.nBlocks) {
        return;
    }
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
    const int32_t inLimit = a.srcLen[block];
    const int32_t outLimit = a.dstCap[block];

    int32_t st = 0;
    int32_t eo = 0;  // 32-bit on purpose: hipcc (ROCm 7.2) mis-merged a 64-bit error offset across the divergent breaks for GS=8/64
    int32_t ip = 0;
    int32_t op = 0;

#define LZ4_FAIL(detail, off)                          \
    {                                                  \
        st = mk_status(ACHIP_CLASS_MALFORMED, detail); \
        eo = (int32_t)(off);                                  \
        break;                                         \
    }

    if (inLimit == 0) {  // :48-50
        st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_LZ4_INPUT_EMPTY);
    }
    else if (outLimit == 0) {  // :52-57 (the Java method returns -1 here)
        if (!(inLimit == 1 && in[0] == 0)) {
            st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT);
        }
    }
    else {
        const int32_t fastOutLimit = outLimit - 8;
        while (ip < inLimit) {
            const int32_t token = in[ip++];

            int32_t lit = token >> 4;  // :62-77
            if (lit == 0xF) {
                if (ip >= inLimit) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                int32_t v;
                do {
                    v = in[ip++];
                    lit = (int32_t)((uint32_t)lit + (uint32_t)v);
                } while (v == 255 && ip < inLimit - 15);
            }
            if (lit < 0) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);

            const int64_t litEnd = (int64_t)ip + lit;
            const int64_t litOutLimit = (int64_t)op + lit;
            if (litOutLimit > fastOutLimit - 4 || litEnd > inLimit - 8) {  // :82-96 last literals
                if (litOutLimit > outLimit) LZ4_FAIL(ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE, ip);
                if (litEnd != inLimit) LZ4_FAIL(ACHIP_D_LZ4_INPUT_NOT_CONSUMED, ip);
                group_copy<GS>(out + op, in + ip, lit, g);
                op += lit;
                break;
            }

            // offset (+ the byte after it) is readable: litEnd <= inLimit - 8
            const uint32_t w = ld4(in + litEnd);
            group_copy<GS>(out + op, in + ip, lit, g);  // :99-109
            op += lit;
            ip = (int32_t)litEnd + 2;

            const int32_t offset = (int32_t)(w & 0xFFFF);  // :113-119
            if (offset == 0 || offset > op) LZ4_FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, ip);

            int32_t ml = token & 0xF;  // :122-138
            if (ml == 0xF) {
                int32_t v;
                bool bad = false;
                do {
                    if (ip > inLimit - 5) {
                        bad = true;
                        break;
                    }
                    v = in[ip++];
                    ml = (int32_t)((uint32_t)ml + (uint32_t)v);
                } while (v == 255);
                if (bad) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
            }
            ml = (int32_t)((uint32_t)ml + 4u);
            if (ml < 0) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);

            const int64_t matchOutLimit = (int64_t)op + ml;
            if (matchOutLimit > fastOutLimit - 4 && matchOutLimit > outLimit - 5) {  // :168-171
                LZ4_FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, ip);
            }
            group_match_copy<GS>(out, op, offset, ml, g);  // :146-194
            op = (int32_t)matchOutLimit;
        }
    }
#undef LZ4_FAIL

    if (g == 0) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

template <int GS>
static hipError_t lz4d_launch_gs(const BatchArgs& a, hipStream_t stream)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    const unsigned grid = (unsigned)((a.nBlocks + GROUPS_PER_WG - 1) / GROUPS_PER_WG);
    hipLaunchKernelGGL(lz4_decompress_kernel<GS>, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_lz4_decompress(const BatchArgs& a, hipStream_t stream, int groupSize)
{
    switch (groupSize) {
        case 1: return lz4d_launch_gs<1>(a, stream);
        case 2: return lz4d_launch_gs<2>(a, stream);
        case 4: return lz4d_launch_gs<4>(a, stream);
        case 16: return lz4d_launch_gs<16>(a, stream);
        case 32: return lz4d_launch_gs<32>(a, stream);
        case 64: return lz4d_launch_gs<64>(a, stream);
        default: return lz4d_launch_gs<8>(a, stream);
    }
}

}  // namespace achip
