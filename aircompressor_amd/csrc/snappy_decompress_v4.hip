// snappy_decompress_v4.hip -- batched Snappy raw-format decode for gfx950: a lane per block with the block's recent output in an LDS
// window (the Snappy counterpart of lz4_decompress_v6.hip; the parse is snappy_decompress_v3.hip's).
//
// Same contract and Java-order checks as snappy_decompress_v2.hip (M/snappy/SnappyRawDecompressor.java:35-322).  Output is appended to a
// 256-byte LDS ring column per lane and leaves for the output buffer in aligned 64-byte pieces; a copy of up to 224 bytes back is read
// from the ring, a farther one from the flushed part of the output buffer.  No cross-lane operation: the test suite also runs the kernel
// on a CPU, one lane at a time (tools/hostemu).
#include "achip_lanewindow.h"

namespace achip {

__device__ __forceinline__ int32_t snappy_op_entry4(int32_t op)  // opLookupTable layout :223-271
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

template <int IN_DW, int OUT_DW>
__global__ __launch_bounds__(64) void snappy_decompress_lanewindow_kernel(BatchArgs a, const int32_t* stats)
{
    using namespace sp;
    if (stats != nullptr && snappy_pick(stats, batch_count(a)) != LZ4_PICK_LANEWINDOW) {  // auto mode: another decoder takes this batch
        return;
    }
    __shared__ uint32_t ldsIn[IN_DW * 64];
    __shared__ uint32_t ldsOut[OUT_DW * 64];
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    const bool have = block < batch_count(a);
    const uint8_t* in0 = have ? a.srcBase + a.srcOff[block] : a.srcBase;
    uint8_t* out = have ? a.dstBase + a.dstOff[block] : a.dstBase;
    const int32_t inLen0 = have ? a.srcLen[block] : 0;
    const int32_t outLimit = have ? a.dstCap[block] : 0;

    int32_t st = 0;
    int32_t eo = 0;
    int32_t op = 0;
    bool done = !have;

    // readUncompressedLength :277-321 (at most 5 bytes: read straight from the input buffer)
    uint32_t expected = 0;
    int32_t nread = 0;
    if (have) {
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
        if (st != 0) {
            done = true;
        }
    }

    // uncompressAll :70-220 ; offsets relative to the first byte after the varint
    const uint8_t* const in = in0 + (done ? 0 : nread);
    const int32_t inLimit = done ? 0 : inLen0 - nread;
    const int32_t fastOutLimit = outLimit - 8;
    int32_t ip = 0;
    LaneInput<IN_DW> R;
    R.init(ldsIn + lane, in, inLimit);
    LaneOutput<OUT_DW> W2;
    W2.init(ldsOut + lane, out);

#define SN_FAIL(off)                                                     \
    {                                                                    \
        st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_MALFORMED); \
        eo = (int32_t)(off);                                             \
        done = true;                                                     \
    }

    int32_t litPos = 0, litOut = 0, litRem = 0;
    int32_t cur = 0, rem = 0, dist = 0;
    int32_t periodic = 0;
    const uint8_t* const inEnd = in + inLimit;
    // bound of the one-store short runs: the announced length, not the capacity -- a caller may hand out capacities that reach into
    // the next block's output (the framed reader does: the Java reader's buffer is larger than a chunk's plaintext)
    const uint8_t* const outEnd = out + (done ? 0 : (int32_t)expected);
    while (!done || rem > 0 || litRem > 0) {  // (lane-private: no cross-lane operation anywhere in this kernel)
        HeadRegs h0;
        h0.A = u32x4{0, 0, 0, 0};
        h0.B = h0.A;
        bool have0 = false;
        if (rem == 0 && litRem == 0 && !done) {
            if (ip >= inLimit) {
                done = true;
            }
            else {
                R.ensure_input(ip, 20);
                const u32x4 W = R.in_u128(ip);
                // one element: tag byte at ip, `t4` the four bytes behind it.  Returns 1 after a literal, 2 after a copy, 0 otherwise
                // (end or failure).  The checks and their order: uncompressAll :84-216.
                auto element = [&](int32_t opc, uint32_t t4) -> int {
                    ip++;
                    const int32_t entry = snappy_op_entry4(opc);
                    const int32_t trailerBytes = entry >> 11;
                    if (!(ip + 4 < inLimit)) {  // :90-92
                        if (ip + trailerBytes > inLimit) {
                            SN_FAIL(ip);
                            return 0;
                        }
                    }
                    // little-endian trailer, masked to trailerBytes (bytes past the input end are never selected)
                    const int32_t trailer = trailerBytes == 0 ? 0 : (int32_t)(t4 & (0xFFFFFFFFu >> (32 - 8 * trailerBytes)));
                    if (trailer < 0) {
                        SN_FAIL(ip);
                        return 0;
                    }
                    ip += trailerBytes;
                    const int32_t length = entry & 0xff;
                    if (length == 0) {
                        return 0;
                    }
                    if ((opc & 3) == 0) {  // literal :116-146
                        const int32_t lit = (int32_t)((uint32_t)length + (uint32_t)trailer);
                        if (lit < 0) {
                            SN_FAIL(ip);
                            return 0;
                        }
                        const int64_t litOutLimit = (int64_t)op + lit;
                        if ((litOutLimit > fastOutLimit || (int64_t)ip + lit > inLimit - 8) && (litOutLimit > outLimit || (int64_t)ip + lit > inLimit)) {
                            SN_FAIL(ip);
                            return 0;
                        }
                        litPos = ip;
                        litOut = op;
                        litRem = lit;
                        ip += lit;
                        op += lit;
                        return 1;
                    }
                    // copy :147-216
                    const int32_t matchOffset = (int32_t)((uint32_t)(entry & 0x700) + (uint32_t)trailer);
                    if (matchOffset <= 0 || matchOffset > op || (int64_t)op + length > outLimit) {
                        SN_FAIL(ip);
                        return 0;
                    }
                    cur = op;
                    rem = length;
                    dist = matchOffset;
                    periodic = cur - matchOffset;
                    op += length;
                    return 2;
                };
                const int32_t tag = (int32_t)(W.x & 0xFF);
                const int kind = element(tag, alignbyte_u32(W.y, W.x, 1));
                if (kind == 1 && (tag >> 2) < 60 && litRem <= 15) {  // the run sits in window bytes 1..15
                    h0.A = u32x4{alignbyte_u32(W.y, W.x, 1), alignbyte_u32(W.z, W.y, 1), alignbyte_u32(W.w, W.z, 1), W.w >> 8};
                    h0.B = h0.A;
                    have0 = true;
                    // a copy right behind a run of <= 10 bytes is in the window too (its tag and up to 4 trailer bytes): same trip
                    const uint32_t at = (uint32_t)litRem + 1u;  // window offset of the next tag: 2..11 for runs of 1..10
                    if (litRem <= 10 && ip < inLimit) {
                        const uint32_t d0 = at < 4 ? W.x : (at < 8 ? W.y : W.z);
                        const uint32_t d1 = at < 4 ? W.y : (at < 8 ? W.z : W.w);
                        const uint32_t d2 = at < 4 ? W.z : (at < 8 ? W.w : 0u);
                        const uint32_t sh = at & 3u;
                        const uint32_t lo = alignbyte_u32(d1, d0, sh), hi = alignbyte_u32(d2, d1, sh);  // window bytes at .. at + 7
                        const int32_t tag2 = (int32_t)(lo & 0xFF);
                        if ((tag2 & 3) != 0) {
                            element(tag2, alignbyte_u32(hi, lo, 1));
                        }
                    }
                }
            }
        }
        // ---- copy through the lane's LDS window (as in lz4_decompress_v6.hip) ----
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
                    dist += dist;
                }
            }
            W2.flush_complete();
        }
    }
#undef SN_FAIL
    if (have) {
        if (st == 0 && (int64_t)expected != (int64_t)op) {  // :61-65
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LENGTH_MISMATCH);
            eo = 0;
        }
        if (st == 0) {
            W2.flush_tail();
        }
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

// auto mode: how long are the elements?  1024 sampled blocks, the elements in the first 768 bytes of each (at most 192; a lane per sample;
// tags only), parsed from an LDS copy of the block's head (lz4_decompress_v5.hip has the reason)
__global__ __launch_bounds__(64) void snappy_element_sample_kernel(BatchArgs a, int32_t* stats, int32_t minBlocks)
{
    const int32_t n = batch_count(a);
    if (n < minBlocks || n <= 0) {
        return;
    }
    constexpr int HEAD = 768, STRIDE = HEAD + 4;
    __shared__ __attribute__((aligned(16))) uint8_t heads[64 * STRIDE];
    const int32_t t = blockIdx.x * 64 + threadIdx.x;
    const int64_t block = (int64_t)t * n / 1024;
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    const int32_t inLen = a.srcLen[block];
    uint8_t* const h = heads + threadIdx.x * STRIDE;
    const int32_t inLimit = inLen < HEAD ? inLen : HEAD;
#pragma unroll 4
    for (int32_t p = 0; p < inLimit; p += 16) {
        if (p + 16 <= inLen) {
            const u32x4 v = ld16(in + p);
            __builtin_memcpy(h + p, &v, 16);
        }
        else {
            for (int32_t i = p; i < inLimit; i++) {
                h[i] = in[i];
            }
        }
    }
    int32_t ip = 0, elements = 0;
    int64_t bytes = 0;
    while (ip < inLimit && ip < 5 && (h[ip] & 0x80) != 0) {  // the length preamble
        ip++;
    }
    ip++;
    while (ip < inLimit && elements < 192) {
        const int32_t opc = h[ip++];
        const int32_t entry = snappy_op_entry4(opc);
        const int32_t trailerBytes = entry >> 11;
        if (ip + trailerBytes > inLimit) {
            break;
        }
        uint32_t trailer = 0;
        for (int i = 0; i < trailerBytes; i++) {
            trailer |= (uint32_t)h[ip + i] << (8 * i);
        }
        ip += trailerBytes;
        int64_t length = entry & 0xff;
        if ((opc & 3) == 0) {
            length += trailer;
            ip += (int32_t)(length < (int64_t)(inLimit - ip) ? length : inLimit - ip);
        }
        bytes += length;
        elements++;
    }
    bytes = bytes < 0 || bytes > (1 << 24) ? (1 << 24) : bytes;
    atomicAdd(stats + 1, elements);
    atomicAdd(stats + 2, (int32_t)(bytes >> 2));
}

hipError_t launch_snappy_element_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks)
{
    hipLaunchKernelGGL(snappy_element_sample_kernel, dim3(16), dim3(64), 0, stream, a, stats, minBlocks);
    return hipGetLastError();
}

hipError_t launch_snappy_decompress_lanewindow(const BatchArgs& a, hipStream_t stream, const int32_t* stats)
{
    const unsigned grid = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL((snappy_decompress_lanewindow_kernel<16, 64>), dim3(grid), dim3(64), 0, stream, a, stats);
    return hipGetLastError();
}

}  // namespace achip
