// snappy_decompress_v3.hip -- batched Snappy raw-format decode for gfx950, a lane per block (achip_lanecopy.h).
//
// Same contract and Java-order checks as snappy_decompress_v2.hip (M/snappy/SnappyRawDecompressor.java:35-322).  The
// structure is that of lz4_decompress_v5.hip: every lane owns a block, a trip parses one element (tag byte + trailer, from
// one 16-byte window of the lane's LDS view of its stream) and the wavefront-wide copy step moves what the 64 elements
// ask for -- a literal run (short ones come out of the window itself) or a copy, one period at a time when it overlaps
// itself.  The instruction stream is the same whatever the blocks of a wavefront contain, which is what a batch of mixed
// data needs (profiles/r01_notes.md).
#include "achip_lanecopy.h"

namespace achip {

__device__ __forceinline__ int32_t snappy_op_entry3(int32_t op)  // opLookupTable layout :223-271
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

template <int IN_DW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void snappy_decompress_lanecopy_kernel(BatchArgs a, const int32_t* mixedGroups)
{
    using namespace sp;
    if (mixedGroups != nullptr && !lz4_batch_is_mixed(*mixedGroups, batch_count(a))) {  // auto mode: the ring decoder takes this batch
        return;
    }
    __shared__ uint32_t ldsIn[IN_DW * 64];
    __shared__ CopyScratch S;
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
    while (__ballot(!done || rem > 0 || litRem > 0) != 0) {
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
                    const int32_t entry = snappy_op_entry3(opc);
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
        // ---- copy (as in lz4_decompress_v5.hip; a trip has a literal run, a copy, or a short run and the copy behind it) ----
        const int32_t n0 = (litRem > LONG || litRem < HEAD) ? litRem : HEAD;
        int32_t n1 = rem < dist ? rem : dist;
        n1 = (n1 > LONG || n1 < HEAD) ? n1 : HEAD;
        n1 = (litRem > n0 || dist < n0 + n1) ? 0 : n1;  // the copy waits for the run in front of it, and for bytes of this trip
        copy_step<true>(S, lane, h0, have0, out + litOut, in + litPos, n0, inEnd, out + cur, out + cur - dist, n1, outEnd);
        litOut += n0;
        litPos += n0;
        litRem -= n0;
        cur += n1;
        rem -= n1;
        if (rem > dist && 2 * (int64_t)dist <= (int64_t)(cur - periodic)) {
            dist += dist;
        }
    }
#undef SN_FAIL
    if (have) {
        if (st == 0 && (int64_t)expected != (int64_t)op) {  // :61-65
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LENGTH_MISMATCH);
            eo = 0;
        }
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

hipError_t launch_snappy_decompress_lanecopy(const BatchArgs& a, hipStream_t stream, const int32_t* mixedGroups)
{
    const unsigned grid = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL((snappy_decompress_lanecopy_kernel<16>), dim3(grid), dim3(64), 0, stream, a, mixedGroups);
    return hipGetLastError();
}

}  // namespace achip
