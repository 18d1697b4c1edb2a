// snappy_decode_body.h -- the Snappy raw-format decode of one buffer over a pair of LDS rings (achip_rings.h), shared by the
// batched decoder (snappy_decompress_v2.hip: GS lanes per block) and the x-snappy-framed reader (snappy_frame.hip: one
// wavefront per stream, its chunks one after another).  Checks in the order of M/snappy/SnappyRawDecompressor.java:35-322.
#pragma once
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

// `ldsIn` / `ldsOut` / `ldsStage` (may be null): the group's rings.  On return st / eo hold the status and error offset, op the
// bytes produced (flushed).  All lanes of the group return the same values.
template <int GS, int IN_RING, int OUT_RING, int GPL, int PHASED = 0>
__device__ __forceinline__ void snappy_buffer_decode(uint8_t* ldsIn, uint8_t* ldsOut, uint8_t* ldsStage, const uint8_t* __restrict__ in0, int32_t inLen0, uint8_t* out,
                                                     int32_t outLimit, int g, int32_t& stOut, int32_t& eoOut, int32_t& opOut)
{
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
        Rings<GS, IN_RING, OUT_RING, GPL, PHASED> R;
        R.init(ldsIn, ldsOut, in, inLimit, out, g, ldsStage);

#define SN_FAIL(off)                                                     \
    {                                                                    \
        st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_MALFORMED); \
        eo = (int32_t)(off);                                             \
        break;                                                           \
    }
        // One trip of the loop = [a literal element, if one is next] then [a copy element, if one is next] -- the shape of an LZ4 sequence.
        // The lane groups of a wavefront run in lockstep and a wavefront pays for every path any of its groups takes: with one element
        // of either kind per trip, groups at a literal and groups at a copy made every trip cost both paths for one element each; in
        // this order the groups stay in step on literal / copy / literal / copy streams and a trip moves two elements.  The elements
        // are still taken strictly in stream order with the checks of :83-216 in their order: status and offset are what they were.
        // element header :84-110 at ip (tag already peeked): false = malformed at eo
        auto header = [&](int32_t opc, int32_t& entry, int32_t& trailer) -> bool {
            ip++;
            entry = snappy_op_entry2(opc);
            const int32_t trailerBytes = entry >> 11;
            if (!(ip + 4 < inLimit)) {  // :90-92
                if (ip + trailerBytes > inLimit) {
                    eo = ip;
                    return false;
                }
            }
            // little-endian trailer: one unaligned 4-byte ring read, masked to trailerBytes (bytes past the input end are never selected)
            const uint32_t t = trailerBytes == 0 ? 0u : (R.template ring_ld4<IN_RING>(R.inRing, ip + R.inBase) & (0xFFFFFFFFu >> (32 - 8 * trailerBytes)));
            trailer = (int32_t)t;
            if (trailer < 0) {
                eo = ip;
                return false;
            }
            ip += trailerBytes;
            return true;
        };
        while (ip < inLimit) {
            R.memory_phase(ip, op);  // (PHASED rings: this trip's refill and flushes, all in one place)
            R.ensure_input(ip, 5);
            int32_t opc = (int32_t)R.in_u8(ip);
            if ((opc & 3) == 0) {  // literal :116-146
                int32_t entry, trailer;
                if (!header(opc, entry, trailer)) SN_FAIL(eo);
                const int32_t length = entry & 0xff;
                if (length != 0) {
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
                if (!(ip < inLimit)) {
                    break;
                }
                R.ensure_input(ip, 5);
                opc = (int32_t)R.in_u8(ip);  // what follows the literal: a copy goes in this trip, another literal in the next
            }
            if ((opc & 3) != 0) {  // copy :147-216
                int32_t entry, trailer;
                if (!header(opc, entry, trailer)) SN_FAIL(eo);
                const int32_t length = entry & 0xff;
                if (length != 0) {
                    const int32_t matchOffset = (int32_t)((uint32_t)(entry & 0x700) + (uint32_t)trailer);
                    if (matchOffset <= 0) SN_FAIL(ip);
                    if (matchOffset > op || (int64_t)op + length > outLimit) SN_FAIL(ip);
                    // A match longer than 64 bytes reaches the decoder as several copy elements with one offset (the writers cut at 64 / 60:
                    // SnappyRawCompressor.java:312-345 -- and a copy element cannot say more).  Elements that continue such a match are taken
                    // in the same trip: the same bytes in the same order, one pass through the copy machinery instead of one per element.
                    // Only behind a full-size piece, only a copy with the same offset whose own checks (:84-110, :147-163) all pass here --
                    // anything else is left to the next trip, which takes it exactly as before.
                    int32_t total = length;
                    if (length >= 60) {
                        while (total < 4096 && ip + 5 < inLimit) {
                            R.ensure_input(ip, 5);
                            const int32_t opc2 = (int32_t)R.in_u8(ip);
                            if ((opc2 & 3) == 0) {
                                break;
                            }
                            const int32_t entry2 = snappy_op_entry2(opc2);
                            const int32_t tb2 = entry2 >> 11;
                            const int32_t trailer2 = (int32_t)(R.template ring_ld4<IN_RING>(R.inRing, ip + 1 + R.inBase) & (0xFFFFFFFFu >> (32 - 8 * tb2)));
                            const int32_t length2 = entry2 & 0xff;
                            if (trailer2 < 0 || length2 == 0 || (int32_t)((uint32_t)(entry2 & 0x700) + (uint32_t)trailer2) != matchOffset || (int64_t)op + total + length2 > outLimit) {
                                break;
                            }
                            total += length2;
                            ip += 1 + tb2;
                            if (length2 < 60) {
                                break;
                            }
                        }
                    }
                    R.copy_match(op, matchOffset, total);
                    op += total;
                }
            }
        }
#undef SN_FAIL
        R.flush_all(op);
        if (st == 0 && (int64_t)expected != (int64_t)op) {  // :61-65
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LENGTH_MISMATCH);
            eo = 0;
        }
    }

    stOut = st;
    eoOut = eo;
    opOut = op;
}

}  // namespace achip
