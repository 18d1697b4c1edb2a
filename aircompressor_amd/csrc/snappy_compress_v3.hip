// snappy_compress_v3.hip -- Snappy raw-format encode, variant 3: the two-tier batch-probe encoder (snappy_compress.hip) with ONE memory round
// trip per batch -- an experiment prepared at the end of round 2, byte-identical with variant 2 and the Java encoder on the CPU emulator,
// not yet measured on a GPU and not the default.  As lz4_compress_v3.hip: the sub-block being encoded is staged through a 1 KiB LDS window
// per wavefront (achip_inwindow.h: the probes' bytes, 16 bytes of match extension, short literal runs come from LDS), and every probe
// loads its candidate's 16 bytes at once, so that the winner of a batch knows a match of up to 16 bytes without another load; longer
// matches continue with the wide compares.  Decisions are the serial encoder's (SnappyRawCompressor.java:47-232).
#include "snappy_compress_body.h"
#include "achip_inwindow.h"

namespace achip {

namespace snc {
constexpr int SNW_WIN = 1024;        // the input window of a wavefront (achip_inwindow.h), refilled 512 bytes at a time: four of them beside the 32 KB
                                     // table leave room for four workgroups per CU (2 KiB windows: three)
constexpr int32_t SNW_MAX_K0 = 160;  // batches whose first probe index is beyond this span more than a chunk (skip step > 7): they load from memory
}

__device__ __forceinline__ void snappy_compress_buffer_window(uint16_t* table, uint8_t* win, const uint8_t* __restrict__ in0, int32_t inLen, uint8_t* __restrict__ out, int32_t outCap, int lane,
                                                       int32_t& stOut, int32_t& outputOut)
{
    using namespace snc;
    using namespace inwin;
    int32_t st = 0;
    int32_t output = 0;
    const int64_t bound = 32 + (int64_t)inLen + inLen / 6;
    if ((int64_t)outCap < bound) {
        st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNAPPY_MAX_OUTPUT);
    }
    else {
        {
            uint32_t n = (uint32_t)inLen;
            int32_t nb = n < (1u << 7) ? 1 : (n < (1u << 14) ? 2 : (n < (1u << 21) ? 3 : (n < (1u << 28) ? 4 : 5)));
            if (lane == 0) {
                for (int i = 0; i < nb; i++) {
                    out[i] = (uint8_t)((n >> (7 * i)) | (i + 1 < nb ? 0x80u : 0u));
                }
            }
            output = nb;
        }
        for (int64_t blockAddress = 0; blockAddress < inLen; blockAddress += BLOCK_SIZE) {
            const uint8_t* __restrict__ in = in0 + blockAddress;
            const int32_t blockLimit = (int32_t)((inLen - blockAddress) < BLOCK_SIZE ? (inLen - blockAddress) : BLOCK_SIZE);
            int32_t tableSize = blockLimit <= 1 ? 0 : (int32_t)((0x80000000u >> __builtin_clz((uint32_t)(blockLimit - 1))) << 1);
            tableSize = tableSize < 256 ? 256 : (tableSize > MAX_HASH_TABLE_SIZE ? MAX_HASH_TABLE_SIZE : tableSize);
            wave_mem_order();
            for (int i = lane; i < tableSize; i += 64) {
                table[i] = 0;
            }
            wave_mem_order();
            const int hashBits = 31 - __builtin_clz((uint32_t)tableSize);
            const int32_t shift = 32 - hashBits;
            const int32_t fastInputLimit = blockLimit - INPUT_MARGIN_BYTES;

            int32_t nextEmit = 0;
            int32_t input = 0;
            int32_t lo = 0, hi = 0;  // the window holds this sub-block's bytes [lo, hi)
            if (input <= fastInputLimit) {
                int mode = 0;           // 0: block start (search only), 1: after a copy, 2: search continues
                int32_t scanStart = 1;  // position of probe 0 of the current search
                int32_t k0 = 0;
                for (;;) {
                    int role = 0;  // 0 idle, 1 insert only, 2 probe
                    int32_t pos = 0;
                    int32_t k = -1;
                    if (mode == 0) {
                        role = 2;
                        k = lane;
                    }
                    else if (mode == 1) {
                        if (lane == 0) {
                            role = 1;
                            pos = input - 1;
                        }
                        else if (lane == 1) {
                            role = 2;
                            pos = input;
                        }
                        else {
                            role = 2;
                            k = lane - 2;
                        }
                    }
                    else {
                        role = 2;
                        k = k0 + lane;
                    }
                    bool valid = true;
                    if (k >= 0) {
                        pos = scanStart + snappy_scan_offset(k);
                        valid = pos + ((32 + k) >> 5) <= fastInputLimit;  // the loop condition of :141
                    }
                    const unsigned long long invalidMask = __ballot(role == 2 && !valid);
                    const int firstInvalid = invalidMask ? __builtin_ctzll(invalidMask) : 64;
                    const bool active = role != 0 && lane < firstInvalid;
                    const unsigned long long activeMask = __ballot(active);

                    // the window covers this batch (its probes lie within a chunk's length) or the batch reads from memory
                    const bool useWin = mode != 2 || k0 <= SNW_MAX_K0;
                    if (useWin) {
                        const int32_t first = mode == 1 ? input - 1 : scanStart + snappy_scan_offset(mode == 2 ? k0 : 0);
                        const int32_t last = scanStart + snappy_scan_offset(mode == 2 ? k0 + 63 : 63) + 16;
                        if (cover<SNW_WIN>(win, in, blockLimit, first, last < blockLimit ? last : blockLimit, lo, hi, lane)) {
                            wave_mem_order();
                        }
                    }
                    uint64_t x0 = 0, x1 = 0;
                    uint32_t x = 0;
                    int32_t h = 0;
                    int32_t cand = 0;
                    bool wide = true;  // x1 holds the second eight bytes (from memory only when they lie inside the sub-block)
                    if (active) {
                        if (useWin) {
                            read16<SNW_WIN>(win, pos, x0, x1);
                        }
                        else {
                            x0 = ld8(in + pos);
                            wide = pos + 16 <= blockLimit;
                            x1 = wide ? ld8(in + pos + 8) : 0ull;
                        }
                        x = (uint32_t)x0;
                        h = snappy_hash(x, shift);
                        cand = (int32_t)table[h];
                    }
                    const unsigned long long same = wave_match_any14((uint32_t)h, hashBits, activeMask);
                    const unsigned long long earlier = same & ((1ull << lane) - 1ull);
                    {
                        const bool fromBatch = active && earlier != 0;
                        const int32_t latest = __shfl(pos, fromBatch ? 63 - __builtin_clzll(earlier) : lane);
                        if (fromBatch) {
                            cand = latest;
                        }
                    }
                    // one round of loads per probe: the candidate's 16 bytes (cand < pos <= blockLimit - 15: they lie inside the sub-block)
                    bool hit = false;
                    int32_t fwd = 0;    // equal bytes from pos / cand on, counted over `avail` bytes
                    int32_t avail = 0;
                    if (active && role == 2) {
                        const uint64_t c0 = ld8(in + cand);
                        const uint64_t c1 = ld8(in + cand + 8);
                        hit = (uint32_t)c0 == x;
                        avail = wide ? 16 : 8;
                        fwd = eq_lead(c0, x0);
                        if (fwd == 8 && wide) {
                            fwd += eq_lead(c1, x1);
                        }
                    }
                    const unsigned long long hitMask = __ballot(hit);
                    const int winner = hitMask ? __builtin_ctzll(hitMask) : -1;
                    const int lastWriter = winner >= 0 ? winner : firstInvalid - 1;
                    {
                        const unsigned long long upTo = lastWriter >= 63 ? ~0ull : ((1ull << (lastWriter + 1)) - 1ull);
                        const unsigned long long later = same & upTo & ~((2ull << lane) - 1ull);
                        if (active && lane <= lastWriter && later == 0) {
                            table[h] = (uint16_t)pos;
                        }
                    }
                    wave_mem_order();

                    if (winner < 0) {
                        if (firstInvalid < 64) {
                            break;  // search ran off the end: remaining bytes are a literal (:160-162)
                        }
                        const int32_t probes = mode == 1 ? 62 : 64;
                        k0 = (mode == 2 ? k0 : 0) + probes;
                        mode = 2;
                        continue;
                    }
                    input = __shfl(pos, winner);
                    const int32_t candidate = __shfl(cand, winner);
                    const int32_t wFwd = __shfl(fwd, winner);
                    const int32_t wAvail = __shfl(avail, winner);
                    const bool reprobe = mode == 1 && winner == 1;
                    if (!reprobe) {  // :169-175
                        const int32_t literalLength = input - nextEmit;
                        output += snappy_literal_header(out, output, literalLength, lane);
                        if (literalLength <= 64 && nextEmit >= lo && input <= hi) {
                            if (lane < literalLength) {  // a short run straight from the window: no load to wait for
                                out[output + lane] = read1<SNW_WIN>(win, nextEmit + lane);
                            }
                        }
                        else {
                            group_copy<64>(out + output, in + nextEmit, literalLength, lane);
                        }
                        output += literalLength;
                    }
                    int32_t matched;  // 4 + count(input + 4, candidate + 4, blockLimit) (:186, :235-266)
                    {
                        const int32_t limitLen = blockLimit - input;
                        if (wFwd < wAvail) {
                            matched = wFwd < limitLen ? wFwd : limitLen;
                        }
                        else if (wAvail >= limitLen) {
                            matched = limitLen;
                        }
                        else {
                            matched = wAvail + wave_count(in, input + wAvail, candidate + wAvail, blockLimit, lane);
                        }
                    }
                    output = snappy_emit_copy(out, output, input - candidate, matched, lane);
                    input += matched;
                    nextEmit = input;
                    if (input >= fastInputLimit) {
                        break;  // :194-196
                    }
                    mode = 1;
                    scanStart = input + 1;
                    k0 = 0;
                }
            }
            if (nextEmit < blockLimit) {  // :224-229
                const int32_t literalLength = blockLimit - nextEmit;
                output += snappy_literal_header(out, output, literalLength, lane);
                group_copy<64>(out + output, in + nextEmit, literalLength, lane);
                output += literalLength;
            }
        }
    }
    stOut = st;
    outputOut = output;
}


// four independent, persistent wavefronts around one LDS table, as snappy_compress_tiers_kernel; each has its own input window
__global__ __launch_bounds__(256) void snappy_compress_tiers_window_kernel(BatchArgs a, uint16_t* slabs, int32_t* nextItem)
{
    using namespace snc;
    __shared__ uint16_t ldsTable[MAX_HASH_TABLE_SIZE];
    __shared__ __attribute__((aligned(16))) uint8_t windows[4][inwin::bytes<SNW_WIN>()];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint16_t* const slab = slabs + ((size_t)blockIdx.x * 3 + (wave > 0 ? wave - 1 : 0)) * MAX_HASH_TABLE_SIZE;
    uint8_t* const win = windows[wave];
    for (;;) {
        int32_t block = 0;
        if (lane == 0) {
            block = atomicAdd(nextItem, 1);
        }
        block = __builtin_amdgcn_readfirstlane(block);
        if (block >= a.nBlocks) {
            return;
        }
        const uint8_t* __restrict__ in0 = a.srcBase + a.srcOff[block];
        uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
        int32_t st = 0;
        int32_t output = 0;
        if (wave == 0) {
            snappy_compress_buffer_window(ldsTable, win, in0, a.srcLen[block], out, a.dstCap[block], lane, st, output);
        }
        else {
            snappy_compress_buffer_window(slab, win, in0, a.srcLen[block], out, a.dstCap[block], lane, st, output);
        }
        if (lane == 0) {
            a.outLen[block] = st == 0 ? output : 0;
            a.status[block] = st;
            a.errOffset[block] = 0;
        }
        wave_mem_order();
    }
}

// scratch: as launch_snappy_compress (snappy_compress_scratch_bytes): [counter: 4 KiB][three table slabs per workgroup]
hipError_t launch_snappy_compress_window(const BatchArgs& a, hipStream_t stream, void* scratch)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    constexpr unsigned WORKGROUPS = 256 * 5;  // (= SNC_TIER_WORKGROUPS of snappy_compress.hip: the scratch is sized for it)
    int32_t* counter = (int32_t*)scratch;
    const hipError_t e = hipMemsetAsync(counter, 0, 64, stream);
    if (e != hipSuccess) return e;
    const unsigned need = (unsigned)((a.nBlocks + 3) / 4);
    hipLaunchKernelGGL(snappy_compress_tiers_window_kernel, dim3(need < WORKGROUPS ? need : WORKGROUPS), dim3(256), 0, stream, a, (uint16_t*)((uint8_t*)scratch + 4096), counter);
    return hipGetLastError();
}

}  // namespace achip
