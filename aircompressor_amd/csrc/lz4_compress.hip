// lz4_compress.hip -- batched LZ4 block encode for gfx950, bit-exact with the Java encoder.
//
// Replaces Lz4RawCompressor.compress + hash/count/emitLiteral/emitMatch/emitLastLiteral/
// encodeRunLength/computeTableSize (M/lz4/Lz4RawCompressor.java:50-312).
//
// One wavefront per block.  The greedy parse is inherently serial per block (which
// positions enter the hash table depends on the parse so far, :113-183), so the parse
// state is wave-uniform and the 64 lanes help inside each step:
//   * hash table (4096 entries, positions relative to the block) lives in LDS -- u16
//     entries for blocks <= 64 KiB (8 KiB per wave => 20 waves / CU), i32 otherwise;
//   * `count` compares 64 x 8 bytes per step and resolves the first mismatch with a
//     ballot (:240-267);
//   * literal runs are copied 64 x 16 bytes per step.
// PROBE_BATCH > 1 additionally evaluates the next PROBE_BATCH probe positions of the
// skip-accelerated search loop (:113-138) in parallel, resolving same-hash collisions
// inside the batch so the table evolves exactly as in program order.
#include "lz4_compress_body.h"
#include "lz4_compress_mw.h"

namespace achip {

template <typename TableT>
__global__ __launch_bounds__(64) void lz4_compress_kernel(BatchArgs a, int32_t bothWidths)
{
    using namespace lz4c;
    __shared__ TableT table[MAX_TABLE_SIZE];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const int32_t inLen = a.srcLen[block];
    constexpr bool WIDE = sizeof(TableT) == 4;
    // two launches cover a batch: u16 tables for blocks <= 64 KiB, i32 tables for the rest
    if (WIDE ? (inLen <= 65536) : (inLen > 65536)) {
        if (!bothWidths && lane == 0) {  // the caller's max_src_len_hint was wrong: say so instead of leaving the block's results unwritten
            a.outLen[block] = 0;
            a.status[block] = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
            a.errOffset[block] = 0;
        }
        return;
    }
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
    const int32_t outCap = a.dstCap[block];

    int32_t st = 0;
    int32_t output = 0;

    const int64_t bound = (int64_t)inLen + inLen / 255 + 16;  // :64-67
    if ((uint32_t)inLen > 0x7E000000u) {                     // :83-85
        st = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_LZ4_MAX_INPUT);
    }
    else if ((int64_t)outCap < bound) {  // :87-89
        st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_MAX_OUTPUT);
    }
    else {
        // computeTableSize :304-311
        int32_t tableSize = inLen <= 1 ? 0 : (int32_t)((0x80000000u >> __builtin_clz((uint32_t)(inLen - 1))) << 1);
        tableSize = tableSize < MIN_TABLE_SIZE ? MIN_TABLE_SIZE : (tableSize > MAX_TABLE_SIZE ? MAX_TABLE_SIZE : tableSize);
        for (int i = lane; i < tableSize; i += 64) {
            table[i] = 0;
        }
        __syncthreads();
        const int32_t mask = tableSize - 1;

        const int32_t inputLimit = inLen;
        const int32_t matchFindLimit = inputLimit - MATCH_FIND_LIMIT;
        const int32_t matchLimit = inputLimit - LAST_LITERAL_SIZE;
        int32_t input = 0;
        int32_t anchor = 0;

        if (inLen >= MIN_LENGTH) {
            table[lz4_hash(ld8(in + input), mask)] = (TableT)input;  // every lane stores the same value: no cross-lane hazard
            input++;
            int32_t nextHash = lz4_hash(ld8(in + input), mask);

            bool done = false;
            do {
                int32_t nextInputIndex = input;
                int32_t findMatchAttempts = 1 << SKIP_TRIGGER;
                int32_t step = 1;
                int32_t matchIndex;
                bool exhausted = false;
                for (;;) {  // :120-138
                    const int32_t hash = nextHash;
                    input = nextInputIndex;
                    nextInputIndex += step;
                    step = (int32_t)((uint32_t)(findMatchAttempts++) >> SKIP_TRIGGER);
                    if (nextInputIndex > matchFindLimit) {
                        exhausted = true;
                        break;
                    }
                    matchIndex = (int32_t)table[hash];
                    nextHash = lz4_hash(ld8(in + nextInputIndex), mask);
                    table[hash] = (TableT)input;
                    if (ld4(in + matchIndex) == ld4(in + input) && matchIndex + MAX_DISTANCE >= input) {
                        break;
                    }
                }
                if (exhausted) {
                    break;  // tail literals from anchor
                }

                // catch up :141-144
                while (input > anchor && matchIndex > 0 && in[input - 1] == in[matchIndex - 1]) {
                    --input;
                    --matchIndex;
                }

                int32_t literalLength = input - anchor;
                int32_t tokenPos = output;
                // emitLiteral :194-207 (token byte is written once the match length is known)
                int32_t litPos = tokenPos + lz4_run_length_size(literalLength);
                group_copy<64>(out + litPos, in + anchor, literalLength, lane);
                output = litPos + literalLength;

                for (;;) {  // :147-184
                    const int32_t matchLength = wave_count(in, input + MIN_MATCH, matchIndex + MIN_MATCH, matchLimit, lane);
                    // emitMatch :209-235
                    if (lane == 0) {
                        lz4_write_run_length(out, tokenPos, literalLength, matchLength >= ML_MASK ? ML_MASK : (uint32_t)matchLength);
                        const uint32_t off = (uint32_t)(input - matchIndex);
                        out[output] = (uint8_t)off;
                        out[output + 1] = (uint8_t)(off >> 8);
                        if (matchLength >= ML_MASK) {
                            int32_t o = output + 2;
                            int32_t remaining = matchLength - ML_MASK;
                            while (remaining >= 510) {
                                out[o++] = 255;
                                out[o++] = 255;
                                remaining -= 510;
                            }
                            if (remaining >= 255) {
                                out[o++] = 255;
                                remaining -= 255;
                            }
                            out[o++] = (uint8_t)remaining;
                        }
                    }
                    output += 2;
                    if (matchLength >= ML_MASK) {
                        output += 1 + (matchLength - ML_MASK) / 255;
                    }

                    input += matchLength + MIN_MATCH;
                    anchor = input;
                    if (input > matchFindLimit) {
                        done = true;
                        break;
                    }

                    const int32_t position = input - 2;
                    const int32_t hp = lz4_hash(ld8(in + position), mask);
                    const int32_t hash = lz4_hash(ld8(in + input), mask);
                    table[hp] = (TableT)position;
                    matchIndex = (int32_t)table[hash];
                    table[hash] = (TableT)input;
                    if (matchIndex + MAX_DISTANCE < input || ld4(in + matchIndex) != ld4(in + input)) {
                        input++;
                        nextHash = lz4_hash(ld8(in + input), mask);
                        break;
                    }
                    // go for another match: zero-literal token
                    tokenPos = output++;
                    literalLength = 0;
                }
            } while (!done);
        }
        {  // emitLastLiteral :269-280 (all three exits of the Java method end here)
            const int32_t length = inputLimit - anchor;
            if (lane == 0) {
                lz4_write_run_length(out, output, length, 0);
            }
            output += lz4_run_length_size(length);
            group_copy<64>(out + output, in + anchor, length, lane);
            output += length;
        }
    }

    if (lane == 0) {
        a.outLen[block] = st == 0 ? output : 0;
        a.status[block] = st;
        a.errOffset[block] = 0;
    }
}


// Batch-probe variant: the 64 lanes evaluate the next 64 steps of the Java search loop at once.
//   lane roles after a match (:157-184): lane 0 = the `input - 2` insert, lane 1 = the immediate re-probe at `input`,
//   lanes 2.. = the probes of the search that follows if the re-probe fails (:113-138, skip schedule included).
// Same-hash collisions inside the batch are resolved so that every probe sees exactly the table state the serial
// loop would have produced (a probe's candidate is the latest earlier batch position with the same hash, else the
// table entry); only the entries up to the winning probe are written back, latest position last.
template <typename TableT>
__global__ __launch_bounds__(64) void lz4_compress_batch_kernel(BatchArgs a, int32_t bothWidths)
{
    using namespace lz4c;
    __shared__ TableT table[MAX_TABLE_SIZE];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const int32_t inLen = a.srcLen[block];
    constexpr bool WIDE = sizeof(TableT) == 4;
    if (WIDE ? (inLen <= 65536) : (inLen > 65536)) {
        if (!bothWidths && lane == 0) {  // the caller's max_src_len_hint was wrong: say so instead of leaving the block's results unwritten
            a.outLen[block] = 0;
            a.status[block] = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
            a.errOffset[block] = 0;
        }
        return;
    }
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
    const int32_t outCap = a.dstCap[block];

    int32_t st = 0;
    const int32_t output = lz4_compress_block<TableT>(in, inLen, out, outCap, table, lane, st);
    if (lane == 0) {
        a.outLen[block] = st == 0 ? output : 0;
        a.status[block] = st;
        a.errOffset[block] = 0;
    }
}

// "Many matches per window" variant (lz4_compress_mw.h): the same parse, replayed over 64 consecutive positions held in registers.
template <typename TableT>
__global__ __launch_bounds__(64) void lz4_compress_mw_kernel(BatchArgs a, int32_t bothWidths)
{
    using namespace lz4c;
    __shared__ TableT table[MAX_TABLE_SIZE];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const int32_t inLen = a.srcLen[block];
    constexpr bool WIDE = sizeof(TableT) == 4;
    if (WIDE ? (inLen <= 65536) : (inLen > 65536)) {
        if (!bothWidths && lane == 0) {
            a.outLen[block] = 0;
            a.status[block] = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
            a.errOffset[block] = 0;
        }
        return;
    }
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
    int32_t st = 0;
    const int32_t output = lz4_compress_block_mw<TableT>(in, inLen, out, a.dstCap[block], table, lane, st);
    if (lane == 0) {
        a.outLen[block] = st == 0 ? output : 0;
        a.status[block] = st;
        a.errOffset[block] = 0;
    }
}

// Two tiers (round 6), the arrangement of snappy_compress_tiers_kernel for the 64 KiB blocks of this encoder.  The 8 KiB table allows twenty wavefronts a CU when
// it sits in LDS, the encoder is a serial chain per block whose wavefronts wait 42 % of their cycles (profiles/r06_counters_lz4_compress_corpus_after2.txt), and 72
// vector registers hold 28.  A workgroup here is FIVE wavefronts with their tables in LDS and up to two with theirs in a slab of global memory (8 KiB each: L2-resident)
// -- four workgroups a CU: the twenty LDS chains as before, and eight slower ones beside them.  Persistent; every wavefront draws its next block from a counter, so
// the slower chains simply take fewer.  (First shape tried: a workgroup of one LDS and one memory wavefront -- 14 + 14 chains a CU: corpus 38.9 -> 44.5 GiB/s,
// fragments 103.6 -> 71.6: on data whose searches run through windows the memory tier's table round trips are what a chain is made of.  profiles/r06_notes.md.)
// (Blocks beyond 64 KiB keep lz4_compress_mw_kernel<int32_t>.)
namespace lz4t {
constexpr int LDS_WAVES = 5;
constexpr int MEM_WAVES_MAX = 2;
constexpr int WORKGROUPS = 256 * 4;
}  // namespace lz4t
__global__ __launch_bounds__(64 * (lz4t::LDS_WAVES + lz4t::MEM_WAVES_MAX), 7) void lz4_compress_tiers_kernel(BatchArgs a, int32_t bothWidths, uint16_t* slabs, int32_t* nextItem)
{
    using namespace lz4c;
    __shared__ uint16_t ldsTable[lz4t::LDS_WAVES][MAX_TABLE_SIZE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint16_t* const slab = slabs + ((size_t)blockIdx.x * lz4t::MEM_WAVES_MAX + (wave >= lz4t::LDS_WAVES ? wave - lz4t::LDS_WAVES : 0)) * MAX_TABLE_SIZE;
    for (;;) {
        int32_t unit = 0;
        if (lane == 0) {
            unit = atomicAdd(nextItem, 1);
        }
        unit = __builtin_amdgcn_readfirstlane(unit);
        if (unit >= a.nBlocks) {
            return;
        }
        const int64_t block = unit;
        const int32_t inLen = uni(a.srcLen[block]);
        if (inLen > 65536) {
            if (!bothWidths && lane == 0) {
                a.outLen[block] = 0;
                a.status[block] = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
                a.errOffset[block] = 0;
            }
            continue;
        }
        const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
        uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
        int32_t st = 0;
        int32_t output;
        if (wave < lz4t::LDS_WAVES) {
            output = lz4_compress_block_mw<uint16_t>(in, inLen, out, a.dstCap[block], ldsTable[wave], lane, st);
        }
        else {
            output = lz4_compress_block_mw<uint16_t>(in, inLen, out, a.dstCap[block], slab, lane, st);
        }
        if (lane == 0) {
            a.outLen[block] = st == 0 ? output : 0;
            a.status[block] = st;
            a.errOffset[block] = 0;
        }
        wave_mem_order();
    }
}

// ---- LZ4 frame container, encoder -------------------------------------------------------------
// Replaces Lz4FrameCompression.compress (M/lz4/Lz4FrameCompression.java:96-140): header (magic, FLG = version 01 +
// independent blocks, BD = 4 MiB, xxHash32 header checksum byte), 4 MiB blocks each through the block encoder above into a
// per-wave scratch slab, stored uncompressed when that is not smaller (:117-131), end mark.  One wavefront per item,
// persistent grid; the capacity checks are made where the Java code makes them (writeInt / ensureCapacity).
namespace lz4f {
constexpr int32_t BLOCK_MAX_4MB = 4 * 1024 * 1024;
constexpr int64_t SLAB_BYTES = ((int64_t)BLOCK_MAX_4MB + BLOCK_MAX_4MB / 255 + 16 + 255) & ~(int64_t)255;
constexpr int32_t MAX_WAVES = 2304;  // 9 wavefronts per CU (16 KB of LDS each) x 256 CUs; round 3 ran 512 (2 per CU): 5 GiB/s where the 64 KiB block encoder makes 28
constexpr int32_t MIN_WAVES = 256;
}  // namespace lz4f

__global__ __launch_bounds__(64) void lz4frame_compress_kernel(BatchArgs a, uint8_t* slabs, int32_t* nextItem)
{
    using namespace lz4c;
    __shared__ int32_t table[MAX_TABLE_SIZE];
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    uint8_t* slab = slabs + (size_t)blockIdx.x * lz4f::SLAB_BYTES;
    for (;;) {
        __syncthreads();
        if (lane == 0) {
            item = atomicAdd(nextItem, 1);
        }
        __syncthreads();
        const int32_t block = item;
        if (block >= a.nBlocks) {
            return;
        }
        const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
        uint8_t* out = a.dstBase + a.dstOff[block];
        const int32_t inLen = a.srcLen[block];
        const int64_t outCap = a.dstCap[block];
        int64_t pos = 0;
        bool tooSmall = false;
        // frame header :102-107
        if (outCap < 7) {
            tooSmall = true;
        }
        else {
            if (lane == 0) {
                st4(out, 0x184D2204u);
                out[4] = 0x60;  // FLG_VERSION | FLG_BLOCK_INDEPENDENCE
                out[5] = 0x70;  // BD_4MB
                out[6] = 0x73;  // (xxHash32({0x60, 0x70}) >>> 8) & 0xFF
            }
            pos = 7;
        }
        int32_t ipos = 0;
        while (!tooSmall && ipos < inLen) {
            const int32_t blockLen = inLen - ipos < lz4f::BLOCK_MAX_4MB ? inLen - ipos : lz4f::BLOCK_MAX_4MB;
            int32_t st = 0;
            wave_mem_order();
            const int32_t clen = lz4_compress_block_mw<int32_t>(in + ipos, blockLen, slab, (int32_t)lz4f::SLAB_BYTES, table, lane, st);
            wave_mem_order();
            const bool compressed = st == 0 && clen < blockLen;
            const int32_t payload = compressed ? clen : blockLen;
            if (pos + 4 > outCap || pos + 4 + payload > outCap) {
                tooSmall = true;
                break;
            }
            if (lane == 0) {
                st4(out + pos, compressed ? (uint32_t)clen : ((uint32_t)blockLen | 0x80000000u));
            }
            pos += 4;
            group_copy<64>(out + pos, compressed ? (const uint8_t*)slab : in + ipos, payload, lane);
            pos += payload;
            ipos += blockLen;
            __syncthreads();
        }
        if (!tooSmall) {
            if (pos + 4 > outCap) {
                tooSmall = true;
            }
            else {
                if (lane == 0) {
                    st4(out + pos, 0u);
                }
                pos += 4;
            }
        }
        if (lane == 0) {
            a.outLen[block] = tooSmall ? 0 : (int32_t)pos;
            a.status[block] = tooSmall ? mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4F_MAX_OUTPUT) : 0;
            a.errOffset[block] = 0;
        }
    }
}

// one 4 MiB slab per resident wavefront: `items` items never need more wavefronts than that (9.7 GB for a batch that fills the chip, on a
// 288 GB device; `least`: what the launch can still work with when the device cannot give that)
int64_t lz4frame_compress_scratch_bytes(int32_t items, bool least)
{
    int64_t waves = items < lz4f::MAX_WAVES ? (items > 0 ? items : 1) : lz4f::MAX_WAVES;
    if (least && waves > lz4f::MIN_WAVES) waves = lz4f::MIN_WAVES;
    return 4096 + waves * lz4f::SLAB_BYTES;
}

hipError_t launch_lz4frame_compress(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    int32_t* counter = (int32_t*)scratch;
    hipError_t e = hipMemsetAsync(counter, 0, 64, stream);
    if (e != hipSuccess) return e;
    int64_t waves = (scratchBytes - 4096) / lz4f::SLAB_BYTES;  // a slab per wavefront
    if (waves < 1) return hipErrorUnknown;  // (the caller sizes the scratch: never)
    if (waves > lz4f::MAX_WAVES) waves = lz4f::MAX_WAVES;
    const unsigned grid = (unsigned)(a.nBlocks < waves ? a.nBlocks : waves);
    hipLaunchKernelGGL(lz4frame_compress_kernel, dim3(grid), dim3(64), 0, stream, a, (uint8_t*)scratch + 4096, counter);
    return hipGetLastError();
}

// the two-tier kernel: from batches that fill the LDS tier on (fewer blocks: a wavefront per block, all tables in LDS)
int g_lz4_mem_waves = 1;            // `lz4.compress.mem_waves`: 0 = one wavefront per block, table in LDS (until round 6); 1 / 2 = memory-tier wavefronts beside five LDS ones.
                                    // Measured, 65 536 blocks (profiles/r06_ab_lz4_compress_tiers.txt): corpus 38.8 / **44.1** / 41.3 GiB/s at 0 / 1 / 2, fragments 103.6 / 104.8 / 93.0
int g_lz4_tier_workgroups = 0;      // (0: lz4t::WORKGROUPS; tools/hostemu makes it small)
int g_lz4_tier_min_blocks = 256 * 20;  // `lz4.compress.tier_min_blocks` (what the LDS tier holds at once)
int64_t lz4_compress_scratch_bytes() { return 4096 + (int64_t)lz4t::WORKGROUPS * lz4t::MEM_WAVES_MAX * lz4c::MAX_TABLE_SIZE * 2; }

hipError_t launch_lz4_compress(const BatchArgs& a, hipStream_t stream, int variant, int maxSrcLenHint, void* scratch)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    // maxSrcLenHint: 0 = unknown, else the caller's promise about the largest srcLen in the batch: when it is <= 64 KiB the launch of
    // the wide-table kernel is skipped (a block that breaks the promise gets an INVALID_ARGUMENT status, not silence)
    const int32_t both = maxSrcLenHint == 0 || maxSrcLenHint > 65536;
    if (variant == 4 && g_lz4_mem_waves > 0 && scratch != nullptr && a.nBlocks >= g_lz4_tier_min_blocks) {
        int32_t* counter = (int32_t*)scratch;
        const hipError_t e = hipMemsetAsync(counter, 0, 64, stream);
        if (e != hipSuccess) return e;
        const unsigned perGroup = (unsigned)(lz4t::LDS_WAVES + g_lz4_mem_waves);
        unsigned groups = g_lz4_tier_workgroups > 0 ? (unsigned)g_lz4_tier_workgroups : (unsigned)lz4t::WORKGROUPS;
        groups = groups > (unsigned)lz4t::WORKGROUPS ? (unsigned)lz4t::WORKGROUPS : groups;
        const unsigned need = ((unsigned)a.nBlocks + perGroup - 1) / perGroup;
        hipLaunchKernelGGL(lz4_compress_tiers_kernel, dim3(need < groups ? need : groups), dim3(64 * perGroup), 0, stream, a, both, (uint16_t*)((uint8_t*)scratch + 4096), counter);
        if (both) {
            hipLaunchKernelGGL(lz4_compress_mw_kernel<int32_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
        }
        return hipGetLastError();
    }
    if (variant == 4) {
        hipLaunchKernelGGL(lz4_compress_mw_kernel<uint16_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
        if (both) {
            hipLaunchKernelGGL(lz4_compress_mw_kernel<int32_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
        }
        return hipGetLastError();
    }
    if (variant == 0) {
        hipLaunchKernelGGL(lz4_compress_kernel<uint16_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
    }
    else {
        hipLaunchKernelGGL(lz4_compress_batch_kernel<uint16_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
    }
    if (both) {
        if (variant == 0) {
            hipLaunchKernelGGL(lz4_compress_kernel<int32_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
        }
        else {
            hipLaunchKernelGGL(lz4_compress_batch_kernel<int32_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
        }
    }
    return hipGetLastError();
}

}  // namespace achip
