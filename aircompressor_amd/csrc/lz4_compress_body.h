// lz4_compress_body.h -- the LZ4 block encoder of one wavefront (lz4_compress.hip has the design notes): constants, token / run-length
// helpers and lz4_compress_block.  Shared by the batched block encoder, the LZ4 frame writer (lz4_compress.hip) and the Hadoop block-stream
// writer (hadoop_streams.hip).
#pragma once
#include "achip_device.h"

namespace achip {

namespace lz4c {
constexpr int HASH_LOG = 12;
constexpr int MAX_TABLE_SIZE = 1 << HASH_LOG;
constexpr int MIN_TABLE_SIZE = 16;
constexpr int MIN_MATCH = 4;
constexpr int LAST_LITERAL_SIZE = 5;
constexpr int MATCH_FIND_LIMIT = 12;
constexpr int MIN_LENGTH = 13;
constexpr int ML_MASK = 15;
constexpr int RUN_MASK = 15;
constexpr int MAX_DISTANCE = 65535;
constexpr int SKIP_TRIGGER = 6;
}  // namespace lz4c

__device__ __forceinline__ int32_t lz4_hash(uint64_t v, int32_t mask)  // :50-62
{
    return (int32_t)(((v * 889523592379ULL) >> 28) & (uint64_t)(uint32_t)mask);
}

// encodeRunLength :282-302 ; lane 0 writes, all lanes return the new offset
__device__ __forceinline__ int32_t lz4_run_length_size(int32_t length)
{
    return length >= lz4c::RUN_MASK ? 2 + (length - lz4c::RUN_MASK) / 255 : 1;
}

__device__ __forceinline__ void lz4_write_run_length(uint8_t* out, int32_t o, int32_t length, uint32_t tokenLow)
{
    if (length >= lz4c::RUN_MASK) {
        out[o++] = (uint8_t)((lz4c::RUN_MASK << 4) | tokenLow);
        int32_t remaining = length - lz4c::RUN_MASK;
        while (remaining >= 255) {
            out[o++] = 255;
            remaining -= 255;
        }
        out[o++] = (uint8_t)remaining;
    }
    else {
        out[o++] = (uint8_t)((length << 4) | tokenLow);
    }
}

// ---- wave-wide "match any": for every lane, the mask of lanes whose `key` (low `bits` bits) equals its own --------
// Twelve ballots + a select/and per bit; used to replay the hash-table updates of a batch of probes in program order.
__device__ __forceinline__ unsigned long long wave_match_any(uint32_t key, int bits, unsigned long long active)
{
    unsigned long long eq = active;
    for (int b = 0; b < bits; b++) {
        const bool bit = (key >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        eq &= bit ? m : ~m;
    }
    return eq;
}

// sum of the first m probe advances of one search (:113-125): adv(0) = 1, adv(k) = (63 + k) >> 6 for k >= 1
__device__ __forceinline__ int32_t lz4_scan_offset(int32_t m)
{
    if (m <= 65) {
        return m;
    }
    const int32_t n = 62 + m;
    const int32_t q = n >> 6, r = n & 63;
    return 1 + 32 * q * (q - 1) + q * (r + 1);
}
__device__ __forceinline__ int32_t lz4_scan_advance(int32_t k) { return k == 0 ? 1 : (63 + k) >> 6; }

// The batch-probe encoder of one block (M/lz4/Lz4RawCompressor.java:74-187) by one wavefront; `table` is MAX_TABLE_SIZE
// entries of LDS.  Returns the compressed length; stOut receives 0 or the status.  Shared by the batched block encoder
// below and the LZ4 frame encoder (lz4_frame.hip).
template <typename TableT>
__device__ int32_t lz4_compress_block(const uint8_t* __restrict__ in, int32_t inLen, uint8_t* __restrict__ out, int32_t outCap, TableT* table, int lane, int32_t& stOut)
{
    using namespace lz4c;
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

        if (inLen >= MIN_LENGTH) {
            // mode 0: block start (lane 0 inserts position 0, search starts at 1); mode 1: after a match; mode 2: search continues
            int mode = 0;
            int32_t input = 0;      // mode 1: position right after the last match
            int32_t scanStart = 1;  // position of probe 0 of the current search
            int32_t k0 = 0;         // first probe index of this batch (mode 2)
            int width = 64;          // lanes used by a batch (a narrow first batch of 8 was measured slower on MI355X: profiles/r01_notes.md)
            for (;;) {
                // ---- roles and positions ----
                int role = 0;  // 0 idle, 1 insert only, 2 probe
                int32_t pos = 0;
                int32_t k = -1;  // search probe index (>= 0 for search probes; -1 for the insert / re-probe lanes)
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
                if (lane >= width) {
                    role = 0;
                    k = -1;
                }
                bool valid = true;  // a search probe whose next index passes matchFindLimit ends the block (:127-129)
                if (k >= 0) {
                    pos = scanStart + lz4_scan_offset(k);
                    valid = pos + lz4_scan_advance(k) <= matchFindLimit;
                }
                const unsigned long long invalidMask = __ballot(role == 2 && !valid);
                const int firstInvalid = invalidMask ? __builtin_ctzll(invalidMask) : 64;
                const bool active = role != 0 && lane < firstInvalid;
                const unsigned long long activeMask = __ballot(active);

                // ---- evaluate every probe against the table state it would see ----
                uint64_t x = 0;
                int32_t h = 0;
                int32_t cand = 0;
                if (active) {
                    x = ld8(in + pos);
                    h = lz4_hash(x, mask);
                    cand = (int32_t)table[h];
                }
                const unsigned long long same = wave_match_any((uint32_t)h, hashBits, activeMask);
                const unsigned long long earlier = same & ((1ull << lane) - 1ull);
                {  // every lane takes part in the shuffle (a lane outside the branch could not be read from)
                    const bool fromBatch = active && earlier != 0;
                    const int32_t latest = __shfl(pos, fromBatch ? 63 - __builtin_clzll(earlier) : lane);
                    if (fromBatch) {
                        cand = latest;
                    }
                }
                bool hit = false;
                if (active && role == 2) {
                    hit = ld4(in + cand) == (uint32_t)x && cand + MAX_DISTANCE >= pos;
                }
                const unsigned long long hitMask = __ballot(hit);
                const int winner = hitMask ? __builtin_ctzll(hitMask) : -1;
                const int lastWriter = winner >= 0 ? winner : firstInvalid - 1;  // lanes 0..lastWriter update the table
                // ---- write back: the latest position of every hash among lanes 0..lastWriter ----
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
                        break;  // search ran off the end: last literals from anchor
                    }
                    // no match in this batch: the search goes on
                    const int32_t probes = mode == 0 ? width - 1 : (mode == 1 ? width - 2 : width);
                    k0 = (mode == 2 ? k0 : 0) + probes;
                    mode = 2;
                    width = 64;
                    continue;
                }
                input = __shfl(pos, winner);
                int32_t matchIndex = __shfl(cand, winner);
                const bool reprobe = mode == 1 && winner == 1;

                int32_t literalLength = 0;
                int32_t tokenPos;
                if (!reprobe) {
                    // catch up :141-144 -- one lane per candidate byte, first mismatch by ballot
                    int32_t room = input - anchor < matchIndex ? input - anchor : matchIndex;
                    while (room > 0) {
                        const bool eq = lane < room && in[input - 1 - lane] == in[matchIndex - 1 - lane];
                        const unsigned long long ne = ~__ballot(eq);
                        const int run = ne ? __builtin_ctzll(ne) : 64;
                        input -= run;
                        matchIndex -= run;
                        room -= run;
                        if (run < 64) {
                            break;
                        }
                    }
                    literalLength = input - anchor;
                    tokenPos = output;
                    const int32_t litPos = tokenPos + lz4_run_length_size(literalLength);
                    group_copy<64>(out + litPos, in + anchor, literalLength, lane);  // emitLiteral :194-207
                    output = litPos + literalLength;
                }
                else {
                    tokenPos = output++;  // zero-literal token :181-183
                }
                const int32_t matchLength = wave_count(in, input + MIN_MATCH, matchIndex + MIN_MATCH, matchLimit, lane);
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
                width = 64;
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

}  // namespace achip
