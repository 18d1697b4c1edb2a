// lz4_decompress_v4.hip -- batched LZ4 block decode for gfx950: lane groups + LDS rings (achip_rings.h) driven as a
// state machine whose every iteration is the same straight-line step for every block of the wavefront.
//
// Same contract and the same Java-order checks as lz4_decompress_v2.hip (M/lz4/Lz4RawDecompressor.java:35-198).  The v2
// loop handles one whole sequence per trip, and the 16 blocks of a wavefront take different branches inside it (length
// extensions, short / long copies, near / far sources, refills and flushes at different moments); a wavefront executes
// the union of what its blocks take -- measured on text: 2.3x the instructions of a wavefront whose blocks are identical.
// Here a trip is:  [token due?  parse]  [match header due?  parse]  [pick the source]  ONE generic <= 64-byte move  [flush]
// whatever the block is doing, so the instruction stream is (almost) the same for all blocks; a sequence takes two trips.
#include "achip_rings.h"

namespace achip {

template <int GS, int IN_RING, int OUT_RING>
__global__ __launch_bounds__(256) void lz4_decompress_steps_kernel(BatchArgs a)
{
    ACHIP_DYNAMIC_LDS(smem);
    using R_t = Rings<GS, IN_RING, OUT_RING, 1>;
    constexpr int CHUNK = R_t::CHUNK;
    constexpr int SLOT = IN_RING + OUT_RING + CHUNK + 16;  // rings, far-match staging area, bank-spreading pad
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

    R_t R;
    uint8_t* slot = smem + grp * SLOT;
    R.init(slot, slot + IN_RING, in, inLimit, out, g, slot + IN_RING + OUT_RING);

    int32_t st = 0;
    int32_t eo = 0;
    int32_t ip = 0;
    int32_t op = 0;
    enum { TOKEN = 0, LITERALS = 1, HEADER = 2, MATCH = 3, DONE = 4 };
    int mode = TOKEN;
    int32_t rem = 0;      // bytes left in the copy in progress
    int32_t dist = 0;     // back-reference distance in use (the offset, doubled while it is shorter than a chunk)
    int32_t token = 0;
    bool lastLiterals = false;
    const int32_t fastOutLimit = outLimit - 8;

#define LZ4_FAIL(detail, off)                          \
    {                                                  \
        st = mk_status(ACHIP_CLASS_MALFORMED, detail); \
        eo = (int32_t)(off);                           \
        mode = DONE;                                   \
    }

    if (inLimit == 0) {  // :48-50
        st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_LZ4_INPUT_EMPTY);
        mode = DONE;
    }
    else if (outLimit == 0) {  // :52-57 (the Java method returns -1 here)
        if (!(inLimit == 1 && in[0] == 0)) {
            st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT);
        }
        mode = DONE;
    }

    while (mode != DONE) {
        if (mode == TOKEN) {
            if (ip >= inLimit) {  // the Java loop condition :59
                mode = DONE;
            }
            else {
                R.ensure_input(ip, 4);
                token = (int32_t)R.in_u8(ip++);
                int32_t lit = token >> 4;  // :62-77
                bool failed = false;
                if (lit == 0xF) {
                    if (ip >= inLimit) {
                        LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                        failed = true;
                    }
                    else {
                        int32_t v;
                        do {
                            R.ensure_input(ip, 1);
                            v = (int32_t)R.in_u8(ip++);
                            lit = (int32_t)((uint32_t)lit + (uint32_t)v);
                        } while (v == 255 && ip < inLimit - 15);
                    }
                }
                if (!failed && lit < 0) {
                    LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                    failed = true;
                }
                if (!failed) {
                    const int64_t litEnd = (int64_t)ip + lit;
                    const int64_t litOutLimit = (int64_t)op + lit;
                    lastLiterals = false;
                    if (litOutLimit > fastOutLimit - 4 || litEnd > inLimit - 8) {  // :82-96 last literals
                        if (litOutLimit > outLimit) {
                            LZ4_FAIL(ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE, ip);
                        }
                        else if (litEnd != inLimit) {
                            LZ4_FAIL(ACHIP_D_LZ4_INPUT_NOT_CONSUMED, ip);
                        }
                        else {
                            lastLiterals = true;
                        }
                    }
                    if (mode != DONE) {
                        rem = lit;
                        mode = lit > 0 ? LITERALS : (lastLiterals ? DONE : HEADER);
                    }
                }
            }
        }
        if (mode == HEADER) {
            R.ensure_input(ip, 3);
            const int32_t offset = (int32_t)(R.in_u8(ip) | (R.in_u8(ip + 1) << 8));  // :113-119
            ip += 2;
            if (offset == 0 || offset > op) {
                LZ4_FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, ip);
            }
            else {
                int32_t ml = token & 0xF;  // :122-138
                bool bad = false;
                if (ml == 0xF) {
                    int32_t v;
                    do {
                        if (ip > inLimit - 5) {
                            bad = true;
                            break;
                        }
                        R.ensure_input(ip, 1);
                        v = (int32_t)R.in_u8(ip++);
                        ml = (int32_t)((uint32_t)ml + (uint32_t)v);
                    } while (v == 255);
                }
                ml = (int32_t)((uint32_t)ml + 4u);
                if (bad || ml < 0) {
                    LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                }
                else {
                    const int64_t matchOutLimit = (int64_t)op + ml;
                    if (matchOutLimit > fastOutLimit - 4 && matchOutLimit > outLimit - 5) {  // :168-171
                        LZ4_FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, ip);
                    }
                    else {
                        rem = ml;
                        dist = offset;
                        mode = MATCH;
                    }
                }
            }
        }
        if (mode == LITERALS || mode == MATCH) {
            // ---- one generic move: c bytes from (src, mask, sV) to the output position ----
            const bool isMatch = mode == MATCH;
            int32_t c = rem < CHUNK ? rem : CHUNK;
            c = (isMatch && dist < c) ? dist : c;  // a trip never reads what it writes
            const uint8_t* src;
            int32_t mask, sV;
            if (!isMatch) {
                R.ensure_input(ip, c);
                src = R.inRing;
                mask = IN_RING - 1;
                sV = ip + R.inBase;
            }
            else if (dist <= R_t::LDS_REACH) {
                src = R.outRing;
                mask = OUT_RING - 1;
                sV = op + R.outBase - dist;
            }
            else {
                // left the LDS window: flushed long ago (dist > LDS_REACH >= 2 * CHUNK); CHUNK source bytes land in the
                // staging area, 16 per lane -- reading past c stays inside this block's own output
                wave_mem_order();
                *(u32x4*)(R.stage + 16 * g) = ld16(R.outAligned + R.outBase + (op - dist) + 16 * g);
                src = R.stage;
                mask = CHUNK - 1;
                sV = 0;
            }
            wave_mem_order();
            R.copy_dwords_rt(src, mask, sV, op + R.outBase, c);
            op += c;
            rem -= c;
            if (isMatch) {
                if (dist < CHUNK) {
                    dist += dist;  // one whole period was written (or the match is over): out[x] = out[x - 2 * dist] too
                }
                if (rem == 0) {
                    mode = TOKEN;
                }
            }
            else {
                ip += c;
                if (rem == 0) {
                    mode = lastLiterals ? DONE : HEADER;
                }
            }
            R.flush_complete(op);
        }
    }
#undef LZ4_FAIL
    if (st == 0) {
        R.flush_all(op);
    }
    if (g == 0) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

template <int GS, int IN_RING, int OUT_RING>
static hipError_t lz4d4_launch(const BatchArgs& a, hipStream_t stream)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    constexpr int SLOT = IN_RING + OUT_RING + 16 * GS + 16;
    const unsigned grid = (unsigned)((a.nBlocks + GROUPS_PER_WG - 1) / GROUPS_PER_WG);
    hipLaunchKernelGGL((lz4_decompress_steps_kernel<GS, IN_RING, OUT_RING>), dim3(grid), dim3(256), (size_t)GROUPS_PER_WG * SLOT, stream, a);
    return hipGetLastError();
}

// groupSize 1 exists for the host emulation (sequential lanes are exact only without cross-lane LDS traffic)
hipError_t launch_lz4_decompress_steps(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass)
{
    switch (groupSize) {
        case 1: return lz4d4_launch<1, 64, 128>(a, stream);
        case 2: return ringClass ? lz4d4_launch<2, 128, 256>(a, stream) : lz4d4_launch<2, 64, 128>(a, stream);
        case 8: return ringClass ? lz4d4_launch<8, 512, 1024>(a, stream) : lz4d4_launch<8, 256, 512>(a, stream);
        default: return ringClass ? lz4d4_launch<4, 256, 512>(a, stream) : lz4d4_launch<4, 128, 256>(a, stream);
    }
}

}  // namespace achip
