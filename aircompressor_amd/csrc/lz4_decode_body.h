// lz4_decode_body.h -- the LZ4 block decode loop over a pair of LDS rings (achip_rings.h), shared by the batched block
// decoder (lz4_decompress_v2.hip: GS lanes per block) and the LZ4 frame decoder (lz4_frame.hip: one wavefront per frame,
// its blocks one after another).  Checks in the order and with the thresholds of M/lz4/Lz4RawDecompressor.java:35-198.
#pragma once
#include "achip_rings.h"

namespace achip {

// `R` is initialised on (in, inLimit, out); on return st / eo hold the status and error offset, op the bytes produced
// (the output is flushed).  All lanes of the group return the same values.
template <int GS, int IN_RING, int OUT_RING, int GPL, int PHASED = 0>
__device__ __forceinline__ void lz4_block_decode(Rings<GS, IN_RING, OUT_RING, GPL, PHASED>& R, const uint8_t* __restrict__ in, int32_t inLimit, int32_t outLimit, int32_t& stOut,
                                                 int32_t& eoOut, int32_t& opOut)
{
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
            const bool early = lit + 3 <= Rings<GS, IN_RING, OUT_RING, GPL, PHASED>::CHUNK;
            if (early) {
                R.ensure_input(ip, lit + 3);
                o4 = R.template ring_ld4<IN_RING>(R.inRing, (int32_t)litEnd + R.inBase);
            }
            R.copy_literals(ip, op, lit);  // :99-109
            op += lit;
            ip = (int32_t)litEnd;
            R.memory_phase(ip, op);  // (PHASED rings: the input ring is topped up once per sequence, here between the two copies: at the top of the
                                     // loop -- where the Snappy decoder has it -- it costs this decoder 6 %, here it gains 2 %: profiles/r03_notes.md)

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
    stOut = st;
    eoOut = eo;
    opOut = op;
}

}  // namespace achip
