// snappy_decompress_v2.hip -- batched Snappy raw-format decode for gfx950, LDS-ring version (the default).
//
// Same contract and Java-order checks as snappy_decompress.hip (M/snappy/SnappyRawDecompressor.java:35-322);
// bytes move through the per-block LDS rings of achip_rings.h (input pulled from HBM once, output
// flushed in whole aligned chunks, near back-references served from LDS).
#include "achip_rings.h"

namespace achip {

__device__ __forceinline__ int32_t snappy_op_entry2(int32_t op)  // opLookupTable layout :223-271
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

template <int GS, int IN_RING, int OUT_RING, int GPL>
__global__ __launch_bounds__(256) void snappy_decompress_rings_kernel(BatchArgs a, const int32_t* mixedGroups)
{
    if (mixedGroups != nullptr && lz4_batch_is_mixed(*mixedGroups, a.nBlocks)) {  // auto mode (achip_abi.cpp): the lane-per-block decoder takes this batch
        return;
    }
    ACHIP_DYNAMIC_LDS(smem);
    constexpr int GROUPS_PER_WG = 256 / GS;
    const int g = threadIdx.x & (GS - 1);
    const int grp = threadIdx.x / GS;
    const int64_t block = (int64_t)blockIdx.x * GROUPS_PER_WG + grp;
    if (block >= a.nBlocks) {
        return;
    }
    const uint8_t* __restrict__ in0 = a.srcBase + a.srcOff[block];
    uint8_t* out = a.dstBase + a.dstOff[block];
    const int32_t inLen0 = a.srcLen[block];
    const int32_t outLimit = a.dstCap[block];

    int32_t st = 0;
    int32_t eo = 0;
    int32_t op = 0;

    // readUncompressedLength :277-321 (at most 5 bytes: read straight from HBM)
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
        Rings<GS, IN_RING, OUT_RING, GPL> R;
        R.init(smem + grp * (IN_RING + OUT_RING + a.ringPad), smem + grp * (IN_RING + OUT_RING + a.ringPad) + IN_RING, in, inLimit, out, g,
           a.ringPad >= 16 * GS * GPL ? smem + grp * (IN_RING + OUT_RING + a.ringPad) + IN_RING + OUT_RING : nullptr);

#define SN_FAIL(off)                                                     \
    {                                                                    \
        st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_MALFORMED); \
        eo = (int32_t)(off);                                             \
        break;                                                           \
    }
        while (ip < inLimit) {
            R.ensure_input(ip, 5);
            const int32_t opc = (int32_t)R.in_u8(ip++);
            const int32_t entry = snappy_op_entry2(opc);
            const int32_t trailerBytes = entry >> 11;
            if (!(ip + 4 < inLimit)) {  // :90-92
                if (ip + trailerBytes > inLimit) SN_FAIL(ip);
            }
            // little-endian trailer: one unaligned 4-byte ring read, masked to trailerBytes (bytes past the input end are never selected)
            const uint32_t t = trailerBytes == 0 ? 0u : (R.template ring_ld4<IN_RING>(R.inRing, ip + R.inBase) & (0xFFFFFFFFu >> (32 - 8 * trailerBytes)));
            const int32_t trailer = (int32_t)t;
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
                R.copy_literals(ip, op, lit);
                ip += lit;
                op += lit;
            }
            else {  // copy :147-216
                const int32_t matchOffset = (int32_t)((uint32_t)(entry & 0x700) + (uint32_t)trailer);
                if (matchOffset <= 0) SN_FAIL(ip);
                if (matchOffset > op || (int64_t)op + length > outLimit) SN_FAIL(ip);
                R.copy_match(op, matchOffset, length);
                op += length;
            }
        }
#undef SN_FAIL
        R.flush_all(op);
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

template <int GS, int IN_RING, int OUT_RING, int GPL = 1>
static hipError_t snd2_launch(const BatchArgs& a, hipStream_t stream, const int32_t* mixedGroups)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    const unsigned grid = (unsigned)((a.nBlocks + GROUPS_PER_WG - 1) / GROUPS_PER_WG);
    const size_t lds = (size_t)GROUPS_PER_WG * (IN_RING + OUT_RING + a.ringPad);
    hipLaunchKernelGGL((snappy_decompress_rings_kernel<GS, IN_RING, OUT_RING, GPL>), dim3(grid), dim3(256), lds, stream, a, mixedGroups);
    return hipGetLastError();
}

hipError_t launch_snappy_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups)
{
    switch (groupSize) {
        case 1: return ringClass ? snd2_launch<1, 128, 256, 4>(a, stream, mixedGroups) : snd2_launch<1, 64, 128, 2>(a, stream, mixedGroups);
        case 2: return ringClass ? snd2_launch<2, 128, 256, 2>(a, stream, mixedGroups) : snd2_launch<2, 64, 128, 1>(a, stream, mixedGroups);
        case 4: return ringClass ? snd2_launch<4, 256, 512>(a, stream, mixedGroups) : snd2_launch<4, 128, 256>(a, stream, mixedGroups);
        case 8: return ringClass ? snd2_launch<8, 512, 1024>(a, stream, mixedGroups) : snd2_launch<8, 256, 512>(a, stream, mixedGroups);
        case 32: return ringClass ? snd2_launch<32, 2048, 4096>(a, stream, mixedGroups) : snd2_launch<32, 1024, 2048>(a, stream, mixedGroups);
        case 64: return ringClass ? snd2_launch<64, 4096, 8192>(a, stream, mixedGroups) : snd2_launch<64, 2048, 4096>(a, stream, mixedGroups);
        default: return ringClass ? snd2_launch<16, 1024, 2048>(a, stream, mixedGroups) : snd2_launch<16, 512, 1024>(a, stream, mixedGroups);
    }
}

}  // namespace achip
