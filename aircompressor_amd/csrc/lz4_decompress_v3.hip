// lz4_decompress_v3.hip -- batched LZ4 block decode for gfx950, lane-per-block version (achip_lanes.h).
//
// Same contract and the same Java-order checks as lz4_decompress_v2.hip (M/lz4/Lz4RawDecompressor.java:35-198).
// Every lane decodes its own block; all lanes of a wavefront run the same step:
//     [token due?  parse token + literal length]  [match header due?  parse offset + match length]  [move <= 16 bytes]
// so a text-like sequence (a few literals, a short match) costs two converged steps for 64 blocks at once.
#include "achip_lanes.h"

namespace achip {

template <int IN_DW, int OUT_DW>
__global__ __launch_bounds__(64) void lz4_decompress_lanes_kernel(BatchArgs a)
{
    __shared__ uint32_t lds[(IN_DW + OUT_DW) * 64];
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    if (block >= a.nBlocks) {
        return;
    }
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    uint8_t* out = a.dstBase + a.dstOff[block];
    const int32_t inLimit = a.srcLen[block];
    const int32_t outLimit = a.dstCap[block];

    LaneRings<IN_DW, OUT_DW> R;
    R.init(lds + lane, lds + IN_DW * 64 + lane, in, inLimit, out);

    int32_t st = 0;
    int32_t eo = 0;
    int32_t ip = 0;
    enum { TOKEN = 0, LITERALS = 1, HEADER = 2, MATCH = 3, DONE = 4 };
    int mode = TOKEN;
    int32_t rem = 0;      // bytes left in the copy in progress
    int32_t dist = 0;     // back-reference distance in use (the offset, doubled while it is shorter than a step)
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
                R.ensure_input(ip, 12);
                uint64_t w = R.in_u64(ip);
                token = (int32_t)(w & 0xFF);
                ip++;
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
                            R.ensure_input(ip, 4);
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
                    const int64_t litOutLimit = (int64_t)R.op() + lit;
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
            R.ensure_input(ip, 12);
            const uint64_t w = R.in_u64(ip);
            const int32_t offset = (int32_t)(w & 0xFFFF);  // :113-119
            ip += 2;
            if (offset == 0 || offset > R.op()) {
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
                        R.ensure_input(ip, 4);
                        v = (int32_t)R.in_u8(ip++);
                        ml = (int32_t)((uint32_t)ml + (uint32_t)v);
                    } while (v == 255);
                }
                ml = (int32_t)((uint32_t)ml + 4u);
                if (bad || ml < 0) {
                    LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                }
                else {
                    const int64_t matchOutLimit = (int64_t)R.op() + ml;
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
        if (mode == LITERALS) {
            const int32_t c = rem < 16 ? rem : 16;
            R.ensure_input(ip, c + 8);
            R.copy_literals_step(ip, c);
            ip += c;
            rem -= c;
            if (rem == 0) {
                mode = lastLiterals ? DONE : HEADER;
            }
        }
        else if (mode == MATCH) {
            // out[op + k] = out[op + k - offset], byte-sequential semantics: a step never reads what it writes (c <= dist),
            // and a distance shorter than a step doubles after a full period has been written (the data is periodic).
            int32_t c = rem < 16 ? rem : 16;
            c = c < dist ? c : dist;
            R.copy_match_step(dist, c);
            rem -= c;
            if (dist < 16) {
                dist += dist;
            }
            if (rem == 0) {
                mode = TOKEN;
            }
        }
    }
#undef LZ4_FAIL
    if (st == 0) {
        R.flush_tail();
    }
    a.outLen[block] = st == 0 ? R.op() : 0;
    a.status[block] = st;
    a.errOffset[block] = (int64_t)eo;
}

template <int IN_DW, int OUT_DW>
static hipError_t lz4d3_launch(const BatchArgs& a, hipStream_t stream)
{
    const unsigned grid = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL((lz4_decompress_lanes_kernel<IN_DW, OUT_DW>), dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

// ringClass: 0 = 128 B in / 256 B history per block (24 KiB per wavefront), 1 = 128 B / 512 B (40 KiB)
hipError_t launch_lz4_decompress_lanes(const BatchArgs& a, hipStream_t stream, int ringClass)
{
    return ringClass ? lz4d3_launch<32, 128>(a, stream) : lz4d3_launch<32, 64>(a, stream);
}

}  // namespace achip
