// lz4_decompress_v6.hip -- batched LZ4 block decode for gfx950: a lane per block (as lz4_decompress_v5.hip) with the block's recent
// output in an LDS window.
//
// Same contract and the same Java-order checks as lz4_decompress_v2.hip (M/lz4/Lz4RawDecompressor.java:35-198); the parse is v5's.
// v5 writes every copy straight to the output buffer and reads every back-reference from it, and is bound by the memory traffic that
// causes with 262144 blocks open at once (profiles/r01_notes.md: 16-byte partial-line stores written to HBM 3.6-5.7 times over, a
// 128-byte line fetched per back-reference).  Here a lane keeps the last 256 bytes of its block in an LDS ring column:
//   * bytes are appended to the ring (funnel-shifted whole dwords) and leave for the output buffer in aligned 64-byte pieces -- four
//     16-byte stores to one line, each byte written once;
//   * a back-reference of up to 224 bytes is read from the ring; a farther one from the flushed part of the output buffer.
// Every copy goes 32 bytes per trip through the ring (no wavefront-wide copy steps): the kernel has no cross-lane operation at all,
// which also lets the test suite run it on a CPU one lane at a time (tools/hostemu).
#include "achip_lanewindow.h"

namespace achip {

template <int IN_DW, int OUT_DW>
__global__ __launch_bounds__(64) void lz4_decompress_lanewindow_kernel(BatchArgs a, const int32_t* mixedGroups)
{
    using namespace sp;
    if (mixedGroups != nullptr && lz4_pick(mixedGroups, batch_count(a)) != LZ4_PICK_LANEWINDOW) {  // auto mode: another decoder takes this batch
        return;
    }
    __shared__ uint32_t ldsIn[IN_DW * 64];
    __shared__ uint32_t ldsOut[OUT_DW * 64];
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    const bool have = block < batch_count(a);
    const uint8_t* in = have ? a.srcBase + a.srcOff[block] : a.srcBase;
    uint8_t* out = have ? a.dstBase + a.dstOff[block] : a.dstBase;
    const int32_t inLimit = have ? a.srcLen[block] : 0;
    const int32_t outLimit = have ? a.dstCap[block] : 0;

    LaneInput<IN_DW> R;
    R.init(ldsIn + lane, in, inLimit);
    LaneOutput<OUT_DW> W2;
    W2.init(ldsOut + lane, out);

    int32_t st = 0;
    int32_t eo = 0;
    int32_t ip = 0;
    int32_t op = 0;
    bool done = !have;
    const int32_t fastOutLimit = outLimit - 8;

#define LZ4_FAIL(detail, off)                          \
    {                                                  \
        st = mk_status(ACHIP_CLASS_MALFORMED, detail); \
        eo = (int32_t)(off);                           \
        done = true;                                   \
    }

    if (have) {
        if (inLimit == 0) {  // :48-50
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_LZ4_INPUT_EMPTY);
            done = true;
        }
        else if (outLimit == 0) {  // :52-57 (the Java method returns -1 here)
            if (!(inLimit == 1 && in[0] == 0)) {
                st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT);
            }
            done = true;
        }
    }

    // the copies in progress, as offsets: literal run (litRem bytes left: in[litPos..] -> out[litOut..]) and match (rem
    // bytes left at out[cur..], distance dist -- the offset, doubled with every whole period written)
    int32_t litPos = 0, litOut = 0, litRem = 0;
    int32_t cur = 0, rem = 0, dist = 0;
    int32_t periodic = 0;  // out[periodic .. cur) repeats with the match's offset: dist may be any multiple of it that stays inside
    const uint8_t* const inEnd = in + inLimit;
    const uint8_t* const outEnd = out + outLimit;
    int32_t tokenMl = 0;     // low nibble of the token whose match header is still to be parsed
    bool headerDue = false;  // the literal run was too long for the window: its match header is parsed when the run is copied
    while (!done || rem > 0 || litRem > 0) {  // (lane-private: no cross-lane operation anywhere in this kernel)
        // ---- parse (lanes whose copies are complete): ONE 16-byte window at ip holds the token, a literal run of up to 12
        // bytes, the offset and the first match-length extension byte -- a whole sequence of the common kind.  A longer run
        // is copied from the input buffer over the next trips and its match header is parsed after it. ----
        HeadRegs h0;
        h0.A = u32x4{0, 0, 0, 0};
        h0.B = h0.A;
        bool have0 = false;
        if (rem == 0 && litRem == 0 && !done) {
            if (!headerDue && ip >= inLimit) {  // the Java loop condition :59
                done = true;
            }
            else {
                R.ensure_input(ip, 20);
                const u32x4 W = R.in_u128(ip);
                uint32_t hdr = W.x;  // header bytes: offset (2), first extension byte
                bool parseHeader = headerDue;
                if (!headerDue) {
                    const int32_t token = (int32_t)(W.x & 0xFF);
                    tokenMl = token & 0xF;
                    ip++;
                    int32_t lit = token >> 4;  // :62-77
                    if (lit == 0xF) {
                        if (ip >= inLimit) {
                            LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                        }
                        else {
                            int32_t v = (int32_t)((W.x >> 8) & 0xFF);  // first extension byte: in the window
                            ip++;
                            lit += v;
                            while (v == 255 && ip < inLimit - 15) {
                                R.ensure_input(ip, 4);
                                v = (int32_t)R.in_u8(ip++);
                                lit = (int32_t)((uint32_t)lit + (uint32_t)v);
                            }
                        }
                    }
                    if (!done && lit < 0) {
                        LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                    }
                    bool lastLiterals = false;
                    if (!done) {
                        const int64_t litEnd = (int64_t)ip + lit;
                        const int64_t litOutLimit = (int64_t)op + lit;
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
                    }
                    if (!done) {
                        litPos = ip;
                        litOut = op;
                        litRem = lit;
                        ip += lit;
                        op += lit;
                        cur = op;
                        if (lastLiterals) {
                            done = true;
                        }
                        if (lit <= 12) {  // (then there was no extension byte) the run sits in window bytes 1..12, the header behind it
                            h0.A = u32x4{alignbyte_u32(W.y, W.x, 1), alignbyte_u32(W.z, W.y, 1), alignbyte_u32(W.w, W.z, 1), W.w >> 8};
                            h0.B = h0.A;
                            have0 = lit > 0;
                            const uint32_t at = (uint32_t)lit + 1u;  // 1..13
                            const uint32_t lo = at < 4 ? W.x : (at < 8 ? W.y : (at < 12 ? W.z : W.w));
                            const uint32_t hi = at < 4 ? W.y : (at < 8 ? W.z : W.w);  // (at >= 12: bytes 12..15 are all in W.w, hi unused)
                            hdr = at >= 12 ? (W.w >> (8 * (at - 12))) : (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (at & 3)));
                            parseHeader = !lastLiterals;
                        }
                        else {
                            headerDue = !lastLiterals;
                        }
                    }
                }
                if (parseHeader && !done) {
                    headerDue = false;
                    const int32_t offset = (int32_t)(hdr & 0xFFFF);  // :113-119
                    ip += 2;
                    int32_t ml = 0;
                    if (offset == 0 || offset > op) {
                        LZ4_FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, ip);
                    }
                    else {
                        ml = tokenMl;  // :122-138
                        bool bad = false;
                        if (ml == 0xF) {
                            if (ip > inLimit - 5) {
                                bad = true;
                            }
                            else {
                                int32_t v = (int32_t)((hdr >> 16) & 0xFF);  // first extension byte: in the window
                                ip++;
                                ml += v;
                                while (v == 255) {
                                    if (ip > inLimit - 5) {
                                        bad = true;
                                        break;
                                    }
                                    R.ensure_input(ip, 4);
                                    v = (int32_t)R.in_u8(ip++);
                                    ml = (int32_t)((uint32_t)ml + (uint32_t)v);
                                }
                            }
                        }
                        ml = (int32_t)((uint32_t)ml + 4u);
                        if (bad || ml < 0) {
                            LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                            ml = 0;
                        }
                        else {
                            const int64_t matchOutLimit = (int64_t)op + ml;
                            if (matchOutLimit > fastOutLimit - 4 && matchOutLimit > outLimit - 5) {  // :168-171
                                LZ4_FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, ip);
                                ml = 0;
                            }
                        }
                    }
                    rem = ml;
                    dist = offset;
                    periodic = cur - offset;
                    op += ml;
                }
            }
        }
        // ---- copy through the lane's LDS window: <= 32 bytes of the literal run, then <= 32 bytes of the match (one period at most);
        // what is left continues in the next trips.  The run is appended before the match reads, so a match may start in it. ----
        {
            const int32_t n0 = litRem < 32 ? litRem : 32;
            if (n0 > 0) {
                u32x4 A, B = {0, 0, 0, 0};
                if (have0) {
                    A = h0.A;
                }
                else {
                    A = safe_ld16(in + litPos, inEnd);
                    if (n0 > 16) {
                        B = safe_ld16(in + litPos + 16, inEnd);
                    }
                }
                W2.append(A, n0 < 16 ? n0 : 16);
                if (n0 > 16) {
                    W2.append(B, n0 - 16);
                }
                litPos += n0;
                litRem -= n0;
            }
            int32_t n1 = rem < dist ? rem : dist;
            n1 = n1 < 32 ? n1 : 32;
            n1 = litRem > 0 ? 0 : n1;
            if (n1 > 0) {
                const int32_t sV = W2.opV - dist;
                u32x4 A, B = {0, 0, 0, 0};
                if (dist <= LaneOutput<OUT_DW>::REACH) {
                    A = W2.read16(sV);
                    if (n1 > 16) {
                        B = W2.read16(sV + 16);
                    }
                }
                else {  // flushed long ago
                    A = ld16(W2.outAligned + sV);
                    if (n1 > 16) {
                        B = ld16(W2.outAligned + sV + 16);
                    }
                }
                W2.append(A, n1 < 16 ? n1 : 16);
                if (n1 > 16) {
                    W2.append(B, n1 - 16);
                }
                cur += n1;
                rem -= n1;
                if (rem > dist && 2 * (int64_t)dist <= (int64_t)(cur - periodic)) {
                    dist += dist;  // enough periods are written: out[x] = out[x - 2 * dist] holds as well
                }
            }
            W2.flush_complete();
        }
    }
#undef LZ4_FAIL
    if (have) {
        if (st == 0) {
            W2.flush_tail();
        }
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

hipError_t launch_lz4_decompress_lanewindow(const BatchArgs& a, hipStream_t stream, const int32_t* mixedGroups)
{
    const unsigned grid = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL((lz4_decompress_lanewindow_kernel<16, 64>), dim3(grid), dim3(64), 0, stream, a, mixedGroups);
    return hipGetLastError();
}

}  // namespace achip
