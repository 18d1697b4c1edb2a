// snappy_compress_body.h -- the Snappy raw-format encoder of one buffer by one wavefront (batch-probing form), shared by the
// batched encoder (snappy_compress.hip) and the x-snappy-framed writer (snappy_frame.hip).  Bit-exact with the Java encoder.
#pragma once
#include "achip_device.h"

namespace achip {

namespace snc {
constexpr int BLOCK_SIZE = 1 << 16;
constexpr int INPUT_MARGIN_BYTES = 15;
constexpr int MAX_HASH_TABLE_SIZE = 1 << 14;
}  // namespace snc

__device__ __forceinline__ int32_t snappy_hash(uint32_t v, int32_t shift) { return (int32_t)((v * 0x1e35a7bdu) >> shift); }  // :368-371

// emitLiteralLength :268-298 -- returns the number of header bytes; lane 0 writes them
__device__ __forceinline__ int32_t snappy_literal_header(uint8_t* out, int32_t o, int32_t literalLength, int lane)
{
    const int32_t n = literalLength - 1;
    int32_t bytes = 0;
    if (n >= 60) {
        bytes = n < (1 << 8) ? 1 : (n < (1 << 16) ? 2 : (n < (1 << 24) ? 3 : 4));
    }
    if (lane == 0) {
        if (n < 60) {
            out[o] = (uint8_t)(n << 2);
        }
        else {
            out[o] = (uint8_t)((59 + bytes) << 2);
            for (int i = 0; i < bytes; i++) {
                out[o + 1 + i] = (uint8_t)((uint32_t)n >> (8 * i));
            }
        }
    }
    return 1 + bytes;
}

// emitCopy :312-345 -- lane 0 writes; every lane returns the new output offset
__device__ __forceinline__ int32_t snappy_emit_copy(uint8_t* out, int32_t o, int32_t offset, int32_t matchLength, int lane)
{
    while (matchLength >= 68) {
        if (lane == 0) {
            out[o] = (uint8_t)(2 + ((64 - 1) << 2));
            out[o + 1] = (uint8_t)offset;
            out[o + 2] = (uint8_t)(offset >> 8);
        }
        o += 3;
        matchLength -= 64;
    }
    if (matchLength > 64) {
        if (lane == 0) {
            out[o] = (uint8_t)(2 + ((60 - 1) << 2));
            out[o + 1] = (uint8_t)offset;
            out[o + 2] = (uint8_t)(offset >> 8);
        }
        o += 3;
        matchLength -= 60;
    }
    if (matchLength < 12 && offset < 2048) {
        if (lane == 0) {
            out[o] = (uint8_t)(1 + ((matchLength - 4) << 2) + ((offset >> 8) << 5));
            out[o + 1] = (uint8_t)offset;
        }
        o += 2;
    }
    else {
        if (lane == 0) {
            out[o] = (uint8_t)(2 + ((matchLength - 1) << 2));
            out[o + 1] = (uint8_t)offset;
            out[o + 2] = (uint8_t)(offset >> 8);
        }
        o += 3;
    }
    return o;
}

__device__ __forceinline__ unsigned long long wave_match_any14(uint32_t key, int bits, unsigned long long active)
{
    unsigned long long eq = active;
    for (int b = 0; b < bits; b++) {
        const bool bit = (key >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        eq &= bit ? m : ~m;
    }
    return eq;
}

// sum of the first m advances of one search (:141): adv(t) = (32 + t) >> 5
__device__ __forceinline__ int32_t snappy_scan_offset(int32_t m)
{
    const int32_t n = 31 + m;
    const int32_t q = n >> 5, r = n & 31;
    return 16 * q * (q - 1) + q * (r + 1);
}

// SnappyRawCompressor.compress (M/snappy/SnappyRawCompressor.java:47-232) of one buffer by one wavefront, 64 steps of the
// search loop per wave step ("batch probing": see snappy_compress.hip).  `table`: MAX_HASH_TABLE_SIZE u16 entries in LDS.
// `table` may live in LDS or in global memory; only this wavefront touches it (a wavefront's LDS and vector memory operations are
// performed in program order, so ordering points are compiler-level).  All lanes return the same values.
__device__ __forceinline__ void snappy_compress_buffer(uint16_t* table, const uint8_t* __restrict__ in0, int32_t inLen, uint8_t* __restrict__ out, int32_t outCap, int lane,
                                                       int32_t& stOut, int32_t& outputOut)
{
    using namespace snc;
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
            if (input <= fastInputLimit) {
                int mode = 0;           // 0: block start (search only), 1: after a copy, 2: search continues
                int32_t scanStart = 1;  // position of probe 0 of the current search
                int32_t k0 = 0;
                int width = 64;  // lanes used by a batch (a narrow first batch of 8 was measured slower on MI355X: profiles/r01_notes.md)
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
                    if (lane >= width) {
                        role = 0;
                        k = -1;
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

                    uint32_t x = 0;
                    int32_t h = 0;
                    int32_t cand = 0;
                    if (active) {
                        x = ld4(in + pos);
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
                    bool hit = false;
                    if (active && role == 2) {
                        hit = ld4(in + cand) == x;
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
                        const int32_t probes = mode == 1 ? width - 2 : width;
                        k0 = (mode == 2 ? k0 : 0) + probes;
                        mode = 2;
                        width = 64;
                        continue;
                    }
                    input = __shfl(pos, winner);
                    const int32_t candidate = __shfl(cand, winner);
                    const bool reprobe = mode == 1 && winner == 1;
                    if (!reprobe) {  // :169-175
                        const int32_t literalLength = input - nextEmit;
                        output += snappy_literal_header(out, output, literalLength, lane);
                        group_copy<64>(out + output, in + nextEmit, literalLength, lane);
                        output += literalLength;
                    }
                    const int32_t matched = 4 + wave_count(in, input + 4, candidate + 4, blockLimit, lane);
                    output = snappy_emit_copy(out, output, input - candidate, matched, lane);
                    input += matched;
                    nextEmit = input;
                    if (input >= fastInputLimit) {
                        break;  // :194-196
                    }
                    mode = 1;
                    scanStart = input + 1;
                    k0 = 0;
                    width = 64;
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

}  // namespace achip
