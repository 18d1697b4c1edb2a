// lz4_decompress_v2.hip -- batched LZ4 block decode for gfx950, LDS-ring version (the default).
//
// Same contract and the same Java-order checks as lz4_decompress.hip (M/lz4/Lz4RawDecompressor.java:35-198);
// the difference is where the bytes live: the compressed stream is pulled from HBM once through a
// per-block LDS input ring (coalesced 16-byte granules), the output is produced into a per-block LDS
// history ring and flushed to HBM in whole aligned chunks, and back-references within LDS_REACH are
// served from LDS (achip_rings.h).  HBM traffic per block ~= compressed bytes + plaintext bytes.
#include "achip_rings.h"

namespace achip {

template <int GS, int IN_RING, int OUT_RING, int GPL>
__global__ __launch_bounds__(256) void lz4_decompress_rings_kernel(BatchArgs a)
{
    ACHIP_DYNAMIC_LDS(smem);
    constexpr int GROUPS_PER_WG = 256 / GS;
    const int g = threadIdx.x & (GS - 1);
    const int grp = threadIdx.x / GS;
    const int64_t block = (int64_t)blockIdx.x * GROUPS_PER_WG + grp;
    if (block >= a.nBlocks) {
        return;
    }
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    uint8_t* out = a.dstBase + a.dstOff[block];
    const int32_t inLimit = a.srcLen[block];
    const int32_t outLimit = a.dstCap[block];

    Rings<GS, IN_RING, OUT_RING, GPL> R;
    R.init(smem + grp * (IN_RING + OUT_RING + a.ringPad), smem + grp * (IN_RING + OUT_RING + a.ringPad) + IN_RING, in, inLimit, out, g,
           a.ringPad >= 16 * GS * GPL ? smem + grp * (IN_RING + OUT_RING + a.ringPad) + IN_RING + OUT_RING : nullptr);

    int32_t st = 0;
    int32_t eo = 0;  // 32-bit on purpose (see lz4_decompress.hip)
    int32_t ip = 0;
    int32_t op = 0;

#define LZ4_FAIL(detail, off)                          \
    {                                                  \
        st = mk_status(ACHIP_CLASS_MALFORMED, detail); \
        eo = (int32_t)(off);                           \
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
        // The 4-byte windows at the token and at the offset are read one phase early (the next token's before the match
        // copy, the offset's before the literal copy) so that they travel with that copy's own LDS reads: two dependent
        // LDS round trips fewer per sequence.
        R.ensure_input(ip, 4);
        uint32_t t4 = R.template ring_ld4<IN_RING>(R.inRing, ip + R.inBase);  // token and the 3 bytes after it
        while (ip < inLimit) {
            const int32_t token = (int32_t)(t4 & 0xFF);
            ip++;

            int32_t lit = token >> 4;  // :62-77
            if (lit == 0xF) {
                if (ip >= inLimit) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                int32_t v = (int32_t)((t4 >> 8) & 0xFF);  // first extension byte (resident: bytes past the input read as 0 and are not used)
                ip++;
                lit = (int32_t)((uint32_t)lit + (uint32_t)v);
                while (v == 255 && ip < inLimit - 15) {
                    R.ensure_input(ip, 1);
                    v = (int32_t)R.in_u8(ip++);
                    lit = (int32_t)((uint32_t)lit + (uint32_t)v);
                }
            }
            if (lit < 0) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);

            const int64_t litEnd = (int64_t)ip + lit;
            const int64_t litOutLimit = (int64_t)op + lit;
            if (litOutLimit > fastOutLimit - 4 || litEnd > inLimit - 8) {  // :82-96 last literals
                if (litOutLimit > outLimit) LZ4_FAIL(ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE, ip);
                if (litEnd != inLimit) LZ4_FAIL(ACHIP_D_LZ4_INPUT_NOT_CONSUMED, ip);
                R.copy_literals(ip, op, lit);
                op += lit;
                break;
            }

            uint32_t o4 = 0;
            const bool early = lit + 3 <= Rings<GS, IN_RING, OUT_RING, GPL>::CHUNK;
            if (early) {
                R.ensure_input(ip, lit + 3);
                o4 = R.template ring_ld4<IN_RING>(R.inRing, (int32_t)litEnd + R.inBase);
            }
            R.copy_literals(ip, op, lit);  // :99-109
            op += lit;
            ip = (int32_t)litEnd;

            if (!early) {
                R.ensure_input(ip, 3);
                o4 = R.template ring_ld4<IN_RING>(R.inRing, ip + R.inBase);  // offset and the first length-extension byte
            }
            const int32_t offset = (int32_t)(o4 & 0xFFFF);  // :113-119
            ip += 2;
            if (offset == 0 || offset > op) LZ4_FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, ip);

            int32_t ml = token & 0xF;  // :122-138
            if (ml == 0xF) {
                if (ip > inLimit - 5) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                int32_t v = (int32_t)((o4 >> 16) & 0xFF);
                ip++;
                ml = (int32_t)((uint32_t)ml + (uint32_t)v);
                bool bad = false;
                while (v == 255) {
                    if (ip > inLimit - 5) {
                        bad = true;
                        break;
                    }
                    R.ensure_input(ip, 1);
                    v = (int32_t)R.in_u8(ip++);
                    ml = (int32_t)((uint32_t)ml + (uint32_t)v);
                }
                if (bad) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
            }
            ml = (int32_t)((uint32_t)ml + 4u);
            if (ml < 0) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);

            const int64_t matchOutLimit = (int64_t)op + ml;
            if (matchOutLimit > fastOutLimit - 4 && matchOutLimit > outLimit - 5) {  // :168-171
                LZ4_FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, ip);
            }
            R.ensure_input(ip, 4);
            t4 = R.template ring_ld4<IN_RING>(R.inRing, ip + R.inBase);  // the next token, ahead of the match copy
            R.copy_match(op, offset, ml);  // :146-194
            op = (int32_t)matchOutLimit;
        }
        R.flush_all(op);
    }
#undef LZ4_FAIL

    if (g == 0) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

template <int GS, int IN_RING, int OUT_RING, int GPL = 1>
static hipError_t lz4d2_launch(const BatchArgs& a, hipStream_t stream)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    const unsigned grid = (unsigned)((a.nBlocks + GROUPS_PER_WG - 1) / GROUPS_PER_WG);
    const size_t lds = (size_t)GROUPS_PER_WG * (IN_RING + OUT_RING + a.ringPad);
    hipLaunchKernelGGL((lz4_decompress_rings_kernel<GS, IN_RING, OUT_RING, GPL>), dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}

// ringClass: 0 = compact rings (more blocks per CU), 1 = large rings (longer LDS reach)
hipError_t launch_lz4_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass)
{
    switch (groupSize) {
        case 1: return ringClass ? lz4d2_launch<1, 128, 256, 4>(a, stream) : lz4d2_launch<1, 64, 128, 2>(a, stream);
        case 2: return ringClass ? lz4d2_launch<2, 128, 256, 2>(a, stream) : lz4d2_launch<2, 64, 128, 1>(a, stream);
        case 4: return ringClass ? lz4d2_launch<4, 256, 512>(a, stream) : lz4d2_launch<4, 128, 256>(a, stream);
        case 8: return ringClass ? lz4d2_launch<8, 512, 1024>(a, stream) : lz4d2_launch<8, 256, 512>(a, stream);
        case 32: return ringClass ? lz4d2_launch<32, 2048, 4096>(a, stream) : lz4d2_launch<32, 1024, 2048>(a, stream);
        case 64: return ringClass ? lz4d2_launch<64, 4096, 8192>(a, stream) : lz4d2_launch<64, 2048, 4096>(a, stream);
        default: return ringClass ? lz4d2_launch<16, 1024, 2048>(a, stream) : lz4d2_launch<16, 512, 1024>(a, stream);
    }
}

}  // namespace achip
