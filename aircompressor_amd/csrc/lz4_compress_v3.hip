// lz4_compress_v3.hip -- LZ4 block encode, variant 3: batch probing with ONE memory round trip per batch (an experiment prepared at the end of
// round 2: byte-identical with the batch-probe encoder and the Java encoder on the CPU emulator, not yet measured on a GPU; not the default).
//
// The batch-probe encoder (lz4_compress_body.h, variant 1) pays four dependent memory round trips per sequence on text: the probes' own
// eight bytes, the candidates' four bytes, the bytes of the catch-up, the bytes of the match extension (and the loads of the literal copy
// sit in the same in-order queue).  Here
//   * the input is staged through a 2 KiB LDS WINDOW (lz4w::WIN), refilled 1 KiB at a time (one round trip per KiB of progress): the probes' bytes, the
//     16 bytes a match is extended over and short literal runs are read from LDS;
//   * every probe loads, IN THE SAME ROUND, 16 bytes at its candidate and the 8 bytes before it (and the 8 bytes before its own
//     position): the winner of a batch knows its match length up to 16 bytes and its catch-up up to 8 bytes without another load --
//     nearly every match of text; longer ones continue with the wide compares of variant 1.
// Decisions are the serial encoder's (Lz4RawCompressor.java:74-187): a probe hits iff its four bytes equal the candidate's and the
// distance fits, the table evolves in program order (wave_match_any), catch-up and extension return the same numbers.
#include "lz4_compress_body.h"
#include "achip_inwindow.h"

namespace achip {

namespace lz4w {
constexpr int WIN = 2048;        // the input window of a wavefront (achip_inwindow.h): refilled 1 KiB at a time
constexpr int32_t MAX_K0 = 448;  // batches whose first probe index is beyond this span more than a chunk (skip step > 8): they load from memory
}  // namespace lz4w

template <typename TableT>
__device__ int32_t lz4_compress_block_window(const uint8_t* __restrict__ in, int32_t inLen, uint8_t* __restrict__ out, int32_t outCap, TableT* table, uint8_t* win, int lane, int32_t& stOut)
{
    using namespace lz4c;
    using namespace lz4w;
    using namespace inwin;
    int32_t st = 0;
    int32_t output = 0;
    const int64_t bound = (int64_t)inLen + inLen / 255 + 16;
    if ((uint32_t)inLen > 0x7E000000u) {
        st = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_LZ4_MAX_INPUT);
    }
    else if ((int64_t)outCap < bound) {
        st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_MAX_OUTPUT);
    }
    else {
        int32_t tableSize = inLen <= 1 ? 0 : (int32_t)((0x80000000u >> __builtin_clz((uint32_t)(inLen - 1))) << 1);
        tableSize = tableSize < MIN_TABLE_SIZE ? MIN_TABLE_SIZE : (tableSize > MAX_TABLE_SIZE ? MAX_TABLE_SIZE : tableSize);
        for (int i = lane; i < tableSize; i += 64) {
            table[i] = 0;
        }
        __syncthreads();
        const int32_t mask = tableSize - 1;
        const int hashBits = 32 - __builtin_clz((uint32_t)mask | 1u);
        const int32_t inputLimit = inLen;
        const int32_t matchFindLimit = inputLimit - MATCH_FIND_LIMIT;
        const int32_t matchLimit = inputLimit - LAST_LITERAL_SIZE;
        int32_t anchor = 0;
        int32_t lo = 0, hi = 0;  // the window holds the input's bytes [lo, hi) (hi - lo <= WIN; zeros behind the input's end)

        if (inLen >= MIN_LENGTH) {
            int mode = 0;           // 0: block start; 1: after a match; 2: the search continues (roles as in lz4_compress_block)
            int32_t input = 0;
            int32_t scanStart = 1;
            int32_t k0 = 0;
            for (;;) {
                // ---- roles and positions ----
                int role = 0;  // 0 idle, 1 insert only, 2 probe
                int32_t pos = 0;
                int32_t k = -1;
                if (mode == 0) {
                    if (lane == 0) {
                        role = 1;
                        pos = 0;
                    }
                    else {
                        role = 2;
                        k = lane - 1;
                    }
                }
                else if (mode == 1) {
                    if (lane == 0) {
                        role = 1;
                        pos = input - 2;
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
                    pos = scanStart + lz4_scan_offset(k);
                    valid = pos + lz4_scan_advance(k) <= matchFindLimit;
                }
                const unsigned long long invalidMask = __ballot(role == 2 && !valid);
                const int firstInvalid = invalidMask ? __builtin_ctzll(invalidMask) : 64;
                const bool active = role != 0 && lane < firstInvalid;
                const unsigned long long activeMask = __ballot(active);

                // ---- the window covers this batch (its probes lie within a chunk's length) or the batch reads from memory ----
                const bool useWin = mode != 2 || k0 <= MAX_K0;
                if (useWin) {
                    const int32_t first = mode == 0 ? 0 : (mode == 1 ? input - 2 : scanStart + lz4_scan_offset(k0));
                    const int32_t last = scanStart + lz4_scan_offset(mode == 2 ? k0 + 63 : 62) + 16;
                    const int32_t need = last < inLen ? last : inLen;
                    if (cover<WIN>(win, in, inLen, first, need, lo, hi, lane)) {
                        __syncthreads();
                    }
                }

                // ---- evaluate every probe against the table state it would see ----
                uint64_t x = 0, x1 = 0;
                int32_t h = 0;
                int32_t cand = 0;
                if (active) {
                    if (useWin) {
                        read16<WIN>(win, pos, x, x1);
                    }
                    else {
                        x = ld8(in + pos);
                        x1 = pos + 16 <= inLen ? ld8(in + pos + 8) : 0ull;
                    }
                    h = lz4_hash(x, mask);
                    cand = (int32_t)table[h];
                }
                const unsigned long long same = wave_match_any((uint32_t)h, hashBits, activeMask);
                const unsigned long long earlier = same & ((1ull << lane) - 1ull);
                {
                    const bool fromBatch = active && earlier != 0;
                    const int32_t latest = __shfl(pos, fromBatch ? 63 - __builtin_clzll(earlier) : lane);
                    if (fromBatch) {
                        cand = latest;
                    }
                }
                // one round of loads per probe: the candidate's 16 bytes, the 8 bytes before it, the 8 bytes before the probe's own position
                bool hit = false;
                int32_t fwd = 0;      // equal bytes from pos / cand on, counted over `avail` bytes
                int32_t avail = 0;    // 16, or 8 when the candidate's second word would pass the end of the input
                int32_t back = -1;    // equal bytes right before pos / cand (0..8), -1: not known (one of them is closer than 8 to the block's start)
                if (active && role == 2) {
                    const uint64_t c0 = ld8(in + cand);
                    const bool two = cand + 16 <= inLen && (useWin || pos + 16 <= inLen);
                    const uint64_t c1 = two ? ld8(in + cand + 8) : 0ull;
                    const bool before = cand >= 8;  // (pos > cand)
                    const uint64_t cb = before ? ld8(in + cand - 8) : 0ull;
                    const uint64_t pb = before ? ld8(in + pos - 8) : 0ull;
                    hit = (uint32_t)c0 == (uint32_t)x && cand + MAX_DISTANCE >= pos;
                    avail = two ? 16 : 8;
                    fwd = eq_lead(c0, x);
                    if (fwd == 8 && two) {
                        fwd += eq_lead(c1, x1);
                    }
                    back = before ? eq_trail(cb, pb) : -1;
                }
                const unsigned long long hitMask = __ballot(hit);
                const int winner = hitMask ? __builtin_ctzll(hitMask) : -1;
                const int lastWriter = winner >= 0 ? winner : firstInvalid - 1;
                {
                    const unsigned long long upTo = lastWriter >= 63 ? ~0ull : ((1ull << (lastWriter + 1)) - 1ull);
                    const unsigned long long later = same & upTo & ~((2ull << lane) - 1ull);
                    if (active && lane <= lastWriter && later == 0) {
                        table[h] = (TableT)pos;
                    }
                }
                __syncthreads();

                if (winner < 0) {
                    if (firstInvalid < 64) {
                        break;
                    }
                    const int32_t probes = mode == 0 ? 63 : (mode == 1 ? 62 : 64);
                    k0 = (mode == 2 ? k0 : 0) + probes;
                    mode = 2;
                    continue;
                }
                input = __shfl(pos, winner);
                int32_t matchIndex = __shfl(cand, winner);
                const int32_t wFwd = __shfl(fwd, winner);
                const int32_t wAvail = __shfl(avail, winner);
                const int32_t wBack = __shfl(back, winner);
                const bool reprobe = mode == 1 && winner == 1;

                // forward: equal bytes from the probe's position on, capped at matchLimit (count(), :240-267, started 4 bytes in)
                int32_t total;
                {
                    const int32_t limitLen = matchLimit - input;
                    if (wFwd < wAvail) {
                        total = wFwd < limitLen ? wFwd : limitLen;
                    }
                    else if (wAvail >= limitLen) {
                        total = limitLen;
                    }
                    else {
                        total = wAvail + wave_count(in, input + wAvail, matchIndex + wAvail, matchLimit, lane);
                    }
                }

                int32_t literalLength = 0;
                int32_t tokenPos;
                int32_t caught = 0;
                if (!reprobe) {
                    // catch up :141-144
                    int32_t room = input - anchor < matchIndex ? input - anchor : matchIndex;
                    if (wBack >= 0 && (wBack < 8 || room <= 8)) {
                        caught = wBack < room ? wBack : room;
                    }
                    else {
                        if (wBack == 8) {  // the eight bytes before are equal: the wide compare goes on from there
                            caught = 8;
                            room -= 8;
                        }
                        while (room > 0) {
                            const bool eq = lane < room && in[input - caught - 1 - lane] == in[matchIndex - caught - 1 - lane];
                            const unsigned long long ne = ~__ballot(eq);
                            const int run = ne ? __builtin_ctzll(ne) : 64;
                            caught += run;
                            room -= run;
                            if (run < 64) {
                                break;
                            }
                        }
                    }
                    input -= caught;
                    matchIndex -= caught;
                    literalLength = input - anchor;
                    tokenPos = output;
                    const int32_t litPos = tokenPos + lz4_run_length_size(literalLength);
                    if (literalLength <= 64 && anchor >= lo && input <= hi) {
                        if (lane < literalLength) {  // a short run straight from the window: no load to wait for
                            out[litPos + lane] = read1<WIN>(win, anchor + lane);
                        }
                    }
                    else {
                        group_copy<64>(out + litPos, in + anchor, literalLength, lane);  // emitLiteral :194-207
                    }
                    output = litPos + literalLength;
                }
                else {
                    tokenPos = output++;  // zero-literal token :181-183
                }
                const int32_t matchLength = caught + total - MIN_MATCH;
                if (lane == 0) {  // emitMatch :209-235
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
                    break;  // :152-155
                }
                mode = 1;
                scanStart = input + 1;
                k0 = 0;
            }
        }
        {  // emitLastLiteral :269-280
            const int32_t length = inputLimit - anchor;
            if (lane == 0) {
                lz4_write_run_length(out, output, length, 0);
            }
            output += lz4_run_length_size(length);
            group_copy<64>(out + output, in + anchor, length, lane);
            output += length;
        }
    }
    stOut = st;
    return output;
}

template <typename TableT>
__global__ __launch_bounds__(64) void lz4_compress_window_kernel(BatchArgs a, int32_t bothWidths)
{
    using namespace lz4c;
    __shared__ TableT table[MAX_TABLE_SIZE];
    __shared__ __attribute__((aligned(16))) uint8_t win[inwin::bytes<lz4w::WIN>()];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const int32_t inLen = a.srcLen[block];
    constexpr bool WIDE = sizeof(TableT) == 4;
    if (WIDE ? (inLen <= 65536) : (inLen > 65536)) {  // two launches cover a batch: u16 tables for blocks <= 64 KiB, i32 tables for the rest
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
    const int32_t output = lz4_compress_block_window<TableT>(in, inLen, out, a.dstCap[block], table, win, lane, st);
    if (lane == 0) {
        a.outLen[block] = st == 0 ? output : 0;
        a.status[block] = st;
        a.errOffset[block] = 0;
    }
}

hipError_t launch_lz4_compress_window(const BatchArgs& a, hipStream_t stream, int maxSrcLenHint)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    const int32_t both = maxSrcLenHint == 0 || maxSrcLenHint > 65536;
    hipLaunchKernelGGL(lz4_compress_window_kernel<uint16_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
    if (both) {
        hipLaunchKernelGGL(lz4_compress_window_kernel<int32_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, both);
    }
    return hipGetLastError();
}

}  // namespace achip
