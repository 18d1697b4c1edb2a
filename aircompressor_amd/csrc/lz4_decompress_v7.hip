// lz4_decompress_v7.hip -- batched LZ4 block decode for gfx950 in two passes: parse to records, then a wavefront per block executes them
// (achip_seqexec.h has the design; DESIGN 4c the measurements).
//
// Same contract and the same Java-order checks as the other LZ4 decoders (M/lz4/Lz4RawDecompressor.java:35-198).  Every check of the
// Java loop depends on lengths, offsets and positions only -- never on a decoded byte -- so the PARSE pass alone decides status, error
// offset and output length of a block exactly as the Java decoder would; the execute pass just moves bytes.
//
//   lz4_parse_kernel     a lane per block (64 blocks per wavefront): one sequence per trip from ONE 16-byte window of the lane's LDS
//                        view of its stream (token, first length bytes, offset -- literal bytes are jumped over, never read), the
//                        Java checks in their order, one 8-byte record out.  Records go to 4 KiB chunks claimed from an arena with
//                        one atomic per wavefront and trip; a block whose records do not fit (arena exhausted) is handed to the ring
//                        decoder afterwards (`only` filter).
//   seq_execute_kernel   a wavefront per block: sx::exec_block.
#include "achip_lanecopy.h"
#include "achip_seqexec.h"

namespace achip {

template <int DBG>
__global__ __launch_bounds__(64) void lz4_parse_kernel(BatchArgs a, sx::ArenaHeader* hdr, sx::BlockMeta* meta, int32_t* only, uint64_t* arena, int32_t maxChunks, const int32_t* stats)
{
    if (stats != nullptr && lz4_pick(stats, a.nBlocks) != LZ4_PICK_TWOPASS) {  // auto mode: the ring decoder takes this batch
        return;
    }
    __shared__ uint32_t ldsIn[16 * 64];
    __shared__ uint64_t ldsRec[8 * 64];  // 8 records per lane, flushed as one 64-byte piece
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    const bool have = block < a.nBlocks;
    const uint8_t* in = have ? a.srcBase + a.srcOff[block] : a.srcBase;
    const int32_t inLimit = have ? a.srcLen[block] : 0;
    const int32_t outLimit = have ? a.dstCap[block] : 0;

    using namespace sp;
    LaneInput<16> R;
    R.init(ldsIn + lane, in, inLimit);

    int32_t st = 0;
    int32_t eo = 0;
    int32_t ip = 0;
    int32_t op = 0;
    bool done = !have;
    bool fallback = false;
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

    sx::RecordWriter W;
    W.init(ldsRec + lane);
    int32_t litEndPrev = 0;  // compressed position behind the previous record's literals (for `skip`)

    while (__ballot(!done || W.recFill > 0) != 0) {  // (uniform)
        W.service<DBG>(done, fallback, hdr, arena, maxChunks, lane);
        if (!done) {
            // ---- one sequence (the Java loop body :59-195 without its copies) ----
            uint32_t rLit = 0, rMl = 0, rOff = 0;
            int32_t litStart = 0;
            bool emit = false;
            if (ip >= inLimit) {  // the loop condition :59
                done = true;
            }
            else {
                R.ensure_input(ip, 20);
                const u32x4 W = R.in_u128(ip);
                const int32_t token = (int32_t)(W.x & 0xFF);
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
                    litStart = ip;
                    rLit = (uint32_t)lit;
                    emit = true;
                    ip += lit;
                    op += lit;
                    if (lastLiterals) {
                        done = true;
                    }
                    else {
                        // offset and the first match-length extension byte: in the token's window when the run is short
                        uint32_t hdr4;
                        if (lit <= 12 && (token >> 4) != 0xF) {
                            const uint32_t at = (uint32_t)lit + 1u;  // 1..13
                            const uint32_t lo = at < 4 ? W.x : (at < 8 ? W.y : (at < 12 ? W.z : W.w));
                            const uint32_t hi = at < 4 ? W.y : (at < 8 ? W.z : W.w);
                            hdr4 = at >= 12 ? (W.w >> (8 * (at - 12))) : (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (at & 3)));
                        }
                        else {
                            R.ensure_input(ip, 8);
                            hdr4 = (uint32_t)R.in_u64(ip);
                        }
                        const int32_t offset = (int32_t)(hdr4 & 0xFFFF);  // :113-119
                        ip += 2;
                        if (offset == 0 || offset > op) {
                            LZ4_FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, ip);
                            emit = false;
                        }
                        else {
                            int32_t ml = token & 0xF;  // :122-138
                            bool bad = false;
                            if (ml == 0xF) {
                                if (ip > inLimit - 5) {
                                    bad = true;
                                }
                                else {
                                    int32_t v = (int32_t)((hdr4 >> 16) & 0xFF);  // first extension byte: in the window
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
                                emit = false;
                            }
                            else {
                                const int64_t matchOutLimit = (int64_t)op + ml;
                                if (matchOutLimit > fastOutLimit - 4 && matchOutLimit > outLimit - 5) {  // :168-171
                                    LZ4_FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, ip);
                                    emit = false;
                                }
                                else {
                                    rMl = (uint32_t)ml;
                                    rOff = (uint32_t)offset;
                                    op += ml;
                                }
                            }
                        }
                    }
                }
            }
            // ---- the record (a failed sequence leaves none: the block's output is void anyway) ----
            if (emit) {
                const int32_t skip = litStart - litEndPrev;
                litEndPrev = litStart + (int32_t)rLit;
                if (rLit > (uint32_t)sx::MAX_LEN || rMl > (uint32_t)sx::MAX_LEN || skip > sx::MAX_SKIP) {
                    fallback = true;  // lengths beyond the record fields (blocks of many megabytes): the ring decoder takes the block
                    done = true;
                    W.recFill = 0;
                }
                else {
                    W.put(sx::rec_pack(rLit, rMl, rOff, (uint32_t)skip));
                }
            }
        }
    }
#undef LZ4_FAIL
    if (have) {
        if (fallback) {
            only[block] = 1;
            meta[block].firstChunk = 0;
            meta[block].count = 0;
            atomicAdd(&hdr->fallbackBlocks, 1);
        }
        else {
            only[block] = 0;
            meta[block].firstChunk = W.firstChunk < 0 ? 0 : W.firstChunk;
            meta[block].count = st == 0 ? W.count : 0;
            a.outLen[block] = st == 0 ? op : 0;
            a.status[block] = st;
            a.errOffset[block] = (int64_t)eo;
        }
    }
}

template <bool RING, int DBG = 0>
__global__ __launch_bounds__(64) void seq_execute_kernel(BatchArgs a, const sx::BlockMeta* meta, const uint64_t* arena, const int32_t* stats, int32_t shortLimit)
{
    if (stats != nullptr && lz4_pick(stats, a.nBlocks, shortLimit) != LZ4_PICK_TWOPASS) {  // auto mode: the ring decoder takes this batch
        return;
    }
    __shared__ __attribute__((aligned(16))) uint8_t ring[RING ? sx::WIN + 16 : 16];
    const int64_t block = blockIdx.x;
    const sx::BlockMeta m = meta[block];
    if (m.count <= 0) {
        return;
    }
    if constexpr (RING) {
        sx::exec_block_ring<DBG>(ring, a.srcBase + a.srcOff[block], a.srcLen[block], a.dstBase + a.dstOff[block], a.dstCap[block], arena, m.firstChunk, m.count, (int)threadIdx.x);
    }
    else {
        sx::exec_block(a.srcBase + a.srcOff[block], a.srcLen[block], a.dstBase + a.dstOff[block], a.dstCap[block], arena, m.firstChunk, m.count, (int)threadIdx.x);
    }
}

// scratch: [header 256 B][meta n x 8][only n x 4][arena, 4 KiB aligned]
int64_t lz4_twopass_scratch_bytes(int32_t nBlocks)
{
    const int64_t fixed = 4096 + (((int64_t)nBlocks * 12 + 4095) & ~4095LL);
    int64_t arena = (int64_t)nBlocks * 65536 + (64LL << 20);
    return fixed + arena;
}

// the execute pass (shared with snappy_decompress_v5.hip): a wavefront per block
hipError_t launch_seq_execute(const BatchArgs& a, hipStream_t stream, const sx::BlockMeta* meta, const uint64_t* arena, int execVariant, const int32_t* stats, int32_t shortLimit)
{
    if (execVariant == 0) {
        hipLaunchKernelGGL(seq_execute_kernel<false>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    else if (execVariant == 101) {
        hipLaunchKernelGGL((seq_execute_kernel<true, 1>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    else if (execVariant == 102) {
        hipLaunchKernelGGL((seq_execute_kernel<true, 2>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    else if (execVariant == 103) {
        hipLaunchKernelGGL((seq_execute_kernel<true, 3>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    else if (execVariant == 105) {
        hipLaunchKernelGGL((seq_execute_kernel<true, 5>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    else if (execVariant == 106) {
        hipLaunchKernelGGL((seq_execute_kernel<true, 6>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    else if (execVariant == 107) {
        hipLaunchKernelGGL((seq_execute_kernel<true, 7>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    else if (execVariant == 104) {
        hipLaunchKernelGGL((seq_execute_kernel<true, 4>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    else {
        hipLaunchKernelGGL((seq_execute_kernel<true, 0>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, (const sx::BlockMeta*)meta, (const uint64_t*)arena, stats, shortLimit);
    }
    return hipGetLastError();
}

hipError_t launch_lz4_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);

hipError_t launch_lz4_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats)
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
        hipLaunchKernelGGL(lz4_parse_kernel<1>, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, stream, a, hdr, meta, only, arena, maxChunks, stats);
    }
    else {
        hipLaunchKernelGGL(lz4_parse_kernel<0>, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, stream, a, hdr, meta, only, arena, maxChunks, stats);
    }
    e = launch_seq_execute(a, stream, meta, arena, execVariant, stats, 12);
    if (e != hipSuccess) return e;
    BatchArgs f = a;
    f.only = only;
    f.onlyStats = stats;
    f.onlyShortLimit = 12;
    e = launch_lz4_decompress_rings(f, stream, groupSize, ringClass, nullptr);
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace achip
