// snappy_decompress_v5.hip -- batched Snappy raw-format decode for gfx950 in two passes: parse to records, then a wavefront per block
// executes them (the Snappy counterpart of lz4_decompress_v7.hip; achip_seqexec.h has the design).
//
// Same contract and Java-order checks as snappy_decompress_v2.hip (M/snappy/SnappyRawDecompressor.java:35-322).  As for LZ4, every check
// of the Java loop depends on lengths, offsets and positions only, so the parse pass decides status, error offset and output length.
// A lane per block; a trip parses one element (tag byte + trailer) -- and the copy right behind a short literal run, which sits in the
// same 16-byte window.  Elements become records {literal run, copy} : a run and the copy behind it share one, a copy behind a copy has
// an empty run, a run behind a run an empty copy.  The executor counts compressed positions from the block's first byte: the length
// preamble is part of the first record's `skip`.
#include "achip_lanecopy.h"
#include "achip_seqexec.h"

namespace achip {

__device__ __forceinline__ int32_t snappy_op_entry5(int32_t op)  // opLookupTable layout :223-271
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

template <int DBG>
__global__ __launch_bounds__(64) void snappy_parse_kernel(BatchArgs a, sx::ArenaHeader* hdr, sx::BlockMeta* meta, int32_t* only, uint64_t* arena, int32_t maxChunks, const int32_t* stats)
{
    if (stats != nullptr && snappy_pick(stats, a.nBlocks) != LZ4_PICK_TWOPASS) {  // auto mode: the ring decoder takes this batch
        return;
    }
    using namespace sp;
    __shared__ uint32_t ldsIn[16 * 64];
    __shared__ uint64_t ldsRec[8 * 64];
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    const bool have = block < batch_count(a);
    const uint8_t* in0 = have ? a.srcBase + a.srcOff[block] : a.srcBase;
    const int32_t inLen0 = have ? a.srcLen[block] : 0;
    const int32_t outLimit = have ? a.dstCap[block] : 0;

    int32_t st = 0;
    int32_t eo = 0;
    int32_t op = 0;
    bool done = !have;
    bool fallback = false;

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
    LaneInput<16> R;
    R.init(ldsIn + lane, in, inLimit);

#define SN_FAIL(off)                                                     \
    {                                                                    \
        st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_MALFORMED); \
        eo = (int32_t)(off);                                             \
        done = true;                                                     \
    }

    sx::RecordWriter W;
    W.init(ldsRec + lane);
    int32_t litEndPrev = 0;  // position (counted from in0) behind the previous record's literals
    // the record under construction: a literal run, the copy behind it -- and the copies behind that one as long as they have the same
    // offset (a long match comes as a string of 64-byte copies: one match to the executor)
    int32_t pLit = 0, pStart = 0, pMl = 0, pOff = 0;

    auto flush_pending = [&]() {
        if (pLit == 0 && pMl == 0) {
            return;
        }
        const int32_t skip = pLit > 0 ? pStart - litEndPrev : 0;
        if (pLit > sx::MAX_LEN || pOff > 0xFFFF || skip > sx::MAX_SKIP) {
            fallback = true;  // lengths / offsets beyond the record fields: the ring decoder takes the block
            done = true;
            W.recFill = 0;
        }
        else {
            if (pLit > 0) {
                litEndPrev = pStart + pLit;
            }
            W.put(sx::rec_pack((uint32_t)pLit, (uint32_t)pMl, (uint32_t)pOff, (uint32_t)skip));
        }
        pLit = 0;
        pMl = 0;
        pOff = 0;
    };

    while (__ballot(!done || W.recFill > 0 || pLit > 0 || pMl > 0) != 0) {  // (uniform)
        W.service<DBG>(done, fallback, hdr, arena, maxChunks, lane);
        if (done) {  // the stream ended: its last record (a failed or handed-over block has none)
            if (fallback || st != 0) {
                pLit = 0;
                pMl = 0;
            }
            flush_pending();
        }
        if (!done) {  // (a trip completes at most one record: a run closes the record before it, the copy in the same window joins the run)
            if (ip >= inLimit) {
                done = true;
            }
            else {
                R.ensure_input(ip, 20);
                const u32x4 Wd = R.in_u128(ip);
                // one element: tag byte at ip, `t4` the four bytes behind it.  The checks and their order: uncompressAll :84-216.
                auto element = [&](int32_t opc, uint32_t t4) -> int {
                    ip++;
                    const int32_t entry = snappy_op_entry5(opc);
                    const int32_t trailerBytes = entry >> 11;
                    if (!(ip + 4 < inLimit)) {  // :90-92
                        if (ip + trailerBytes > inLimit) {
                            SN_FAIL(ip);
                            return 0;
                        }
                    }
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
                        flush_pending();  // a run closes the record before it
                        pLit = lit;
                        pStart = nread + ip;
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
                    if (pMl > 0 && (matchOffset != pOff || pMl + length > sx::MAX_LEN)) {
                        flush_pending();
                    }
                    pMl += length;
                    pOff = matchOffset;
                    op += length;
                    return 2;
                };
                const int32_t tag = (int32_t)(Wd.x & 0xFF);
                const int kind = element(tag, alignbyte_u32(Wd.y, Wd.x, 1));
                if (kind == 1 && !done && (tag >> 2) < 60 && pLit <= 10 && ip < inLimit) {
                    // the copy right behind a run of <= 10 bytes is in the window too (its tag and up to 4 trailer bytes): same trip
                    const uint32_t at = (uint32_t)pLit + 1u;  // window offset of the next tag: 2..11
                    const uint32_t d0 = at < 4 ? Wd.x : (at < 8 ? Wd.y : Wd.z);
                    const uint32_t d1 = at < 4 ? Wd.y : (at < 8 ? Wd.z : Wd.w);
                    const uint32_t d2 = at < 4 ? Wd.z : (at < 8 ? Wd.w : 0u);
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
#undef SN_FAIL
    if (!have && block < a.nBlocks) {  // (a batch assembled on the device may hold fewer blocks than the launch was sized for)
        only[block] = 0;
        meta[block].firstChunk = 0;
        meta[block].count = 0;
    }
    if (have) {
        if (fallback) {
            only[block] = 1;
            meta[block].firstChunk = 0;
            meta[block].count = 0;
            atomicAdd(&hdr->fallbackBlocks, 1);
        }
        else {
            if (st == 0 && (int64_t)expected != (int64_t)op) {  // :61-65
                st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LENGTH_MISMATCH);
                eo = 0;
            }
            only[block] = 0;
            meta[block].firstChunk = W.firstChunk < 0 ? 0 : W.firstChunk;
            meta[block].count = st == 0 ? W.count : 0;
            a.outLen[block] = st == 0 ? op : 0;
            a.status[block] = st;
            a.errOffset[block] = (int64_t)eo;
        }
    }
}

hipError_t launch_seq_execute(const BatchArgs& a, hipStream_t stream, const sx::BlockMeta* meta, const uint64_t* arena, int execVariant, const int32_t* stats, int32_t shortLimit);
int64_t lz4_twopass_scratch_bytes(int32_t nBlocks);
hipError_t launch_snappy_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);

int64_t snappy_twopass_scratch_bytes(int32_t nBlocks) { return lz4_twopass_scratch_bytes(nBlocks); }

hipError_t launch_snappy_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* s = (uint8_t*)scratch;
    sx::ArenaHeader* hdr = (sx::ArenaHeader*)s;
    sx::BlockMeta* meta = (sx::BlockMeta*)(s + 4096);
    int32_t* only = (int32_t*)(s + 4096 + (int64_t)a.nBlocks * 8);
    const int64_t fixed = 4096 + (((int64_t)a.nBlocks * 12 + 4095) & ~4095LL);
    uint64_t* arena = (uint64_t*)(s + fixed);
    const int64_t chunks = (scratchBytes - fixed) / (sx::CHUNK_SLOTS * 8);
    const int32_t maxChunks = (int32_t)(chunks > 0x7FFFFFFF ? 0x7FFFFFFF : chunks);
    hipError_t e = hipMemsetAsync(hdr, 0, sizeof(sx::ArenaHeader), stream);
    if (e != hipSuccess) return e;
    if (execVariant == 201) {
        hipLaunchKernelGGL(snappy_parse_kernel<1>, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, stream, a, hdr, meta, only, arena, maxChunks, stats);
    }
    else {
        hipLaunchKernelGGL(snappy_parse_kernel<0>, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, stream, a, hdr, meta, only, arena, maxChunks, stats);
    }
    e = launch_seq_execute(a, stream, meta, arena, execVariant, stats, 6);
    if (e != hipSuccess) return e;
    BatchArgs f = a;
    f.only = only;
    f.onlyStats = stats;
    f.onlyShortLimit = 6;
    e = launch_snappy_decompress_rings(f, stream, groupSize, ringClass, nullptr);
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace achip
