// snappy_decompress.hip -- batched Snappy raw-format decode for gfx950.
//
// Replaces SnappyRawDecompressor.decompress / uncompressAll / readUncompressedLength
// (M/snappy/SnappyRawDecompressor.java:35-322).  Same execution shape as the LZ4 decoder:
// one GS-lane group per block, the element grammar walked with the Java loop's checks in
// the Java loop's order, byte moves 16 bytes per lane.  The 256-entry opLookupTable
// (:227-271) is recomputed from the tag bits (its documented layout) instead of being
// fetched -- two ALU ops beat a dependent LDS/const read on this path.
#include "achip_device.h"

namespace achip {

// entry layout (:223-236): bits 0-7 length, 8-10 copy offset / 256, 11-13 trailer bytes
__device__ __forceinline__ int32_t snappy_op_entry(int32_t op)
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

template <int GS>
__global__ __launch_bounds__(256) void snappy_decompress_kernel(BatchArgs a)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    const int g = threadIdx.x & (GS - 1);
    const int64_t block = (int64_t)blockIdx.x * GROUPS_PER_WG + (threadIdx.x / GS);
    if (block >= a.nBlocks) {
        return;
    }
    const uint8_t* __restrict__ in0 = a.srcBase + a.srcOff[block];
    uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
    const int32_t inLen0 = a.srcLen[block];
    const int32_t outLimit = a.dstCap[block];

    int32_t st = 0;
    int32_t eo = 0;  // 32-bit on purpose: hipcc (ROCm 7.2) mis-merged a 64-bit error offset across the divergent breaks for GS=8/64
    int32_t op = 0;

    // readUncompressedLength :277-321
    uint32_t expected = 0;
    int32_t nread = 0;
    for (int i = 0; i < 5; i++) {
        if (nread >= inLen0) {
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_TRUNCATED);
            eo = inLen0 - nread;
            break;
        }
        const uint32_t b = in0[nread++];
        expected |= (b & 0x7f) << (7 * i);
        if ((b & 0x80) == 0) {
            break;
        }
        if (i == 4) {
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LEN_HIGH_BIT);
            eo = nread;
        }
    }
    if (st == 0 && (int32_t)expected < 0) {
        st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_INVALID_LENGTH);
        eo = 0;
    }
    if (st == 0 && (int64_t)expected > (int64_t)outLimit) {  // :49-50
        st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNAPPY_OUTPUT_TOO_SMALL);
        eo = 0;
    }

    if (st == 0) {
        // uncompressAll :70-220 ; offsets relative to the first byte after the varint
        const uint8_t* __restrict__ in = in0 + nread;
        const int32_t inLimit = inLen0 - nread;
        const int32_t fastOutLimit = outLimit - 8;
        int32_t ip = 0;

#define SN_FAIL(off)                                                       \
    {                                                                      \
        st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_MALFORMED);   \
        eo = (int32_t)(off);                                                      \
        break;                                                             \
    }
        while (ip < inLimit) {
            const int32_t opc = in[ip++];
            const int32_t entry = snappy_op_entry(opc);
            const int32_t trailerBytes = entry >> 11;
            int32_t trailer = 0;
            if (ip + 4 < inLimit) {  // :87-89
                const uint32_t mask = trailerBytes == 0 ? 0u : (0xFFFFFFFFu >> (32 - 8 * trailerBytes));
                trailer = (int32_t)(ld4(in + ip) & mask);
            }
            else {
                if (ip + trailerBytes > inLimit) SN_FAIL(ip);
                uint32_t t = 0;
                for (int k = trailerBytes - 1; k >= 0; k--) {
                    t = (t << 8) | in[ip + k];
                }
                trailer = (int32_t)t;
            }
            if (trailer < 0) SN_FAIL(ip);
            ip += trailerBytes;

            const int32_t length = entry & 0xff;
            if (length == 0) {
                continue;
            }

            if ((opc & 3) == 0) {  // literal :116-146
                const int32_t lit = (int32_t)((uint32_t)length + (uint32_t)trailer);
                if (lit < 0) SN_FAIL(ip);
                const int64_t litOutLimit = (int64_t)op + lit;
                if (litOutLimit > fastOutLimit || (int64_t)ip + lit > inLimit - 8) {
                    if (litOutLimit > outLimit || (int64_t)ip + lit > inLimit) SN_FAIL(ip);
                }
                group_copy<GS>(out + op, in + ip, lit, g);
                ip += lit;
                op += lit;
            }
            else {  // copy :147-216
                const int32_t matchOffset = (int32_t)((uint32_t)(entry & 0x700) + (uint32_t)trailer);
                if (matchOffset <= 0) SN_FAIL(ip);
                if (matchOffset > op || (int64_t)op + length > outLimit) SN_FAIL(ip);
                group_match_copy<GS>(out, op, matchOffset, length, g);
                op += length;
            }
        }
#undef SN_FAIL
        if (st == 0 && (int64_t)expected != (int64_t)op) {  // :61-65
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LENGTH_MISMATCH);
            eo = 0;
        }
    }

    if (g == 0) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

template <int GS>
static hipError_t snd_launch_gs(const BatchArgs& a, hipStream_t stream)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    const unsigned grid = (unsigned)((a.nBlocks + GROUPS_PER_WG - 1) / GROUPS_PER_WG);
    hipLaunchKernelGGL(snappy_decompress_kernel<GS>, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_snappy_decompress(const BatchArgs& a, hipStream_t stream, int groupSize)
{
    switch (groupSize) {
        case 1: return snd_launch_gs<1>(a, stream);
        case 2: return snd_launch_gs<2>(a, stream);
        case 4: return snd_launch_gs<4>(a, stream);
        case 16: return snd_launch_gs<16>(a, stream);
        case 32: return snd_launch_gs<32>(a, stream);
        case 64: return snd_launch_gs<64>(a, stream);
        default: return snd_launch_gs<8>(a, stream);
    }
}

}  // namespace achip
