// snappy_compress.hip -- batched Snappy raw-format encode for gfx950, bit-exact with the Java encoder.
//
// Replaces SnappyRawCompressor.compress + count/emitLiteralLength/fastCopy/emitCopy/
// getHashTableSize/hashBytes/writeUncompressedLength (M/snappy/SnappyRawCompressor.java:47-411).
//
// One wavefront per input buffer; the buffer's independent 64 KiB sub-blocks (:93-99) are
// walked in order because their outputs are concatenated.  The u16 hash table (<= 16384
// entries = 32 KiB) lives in LDS (5 waves / CU).  The greedy parse state is wave-uniform;
// the lanes help with the match-length count (64 x 8 bytes per step, ballot for the first
// mismatch) and the literal copies (64 x 16 bytes per step).
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

__global__ __launch_bounds__(64) void snappy_compress_kernel(BatchArgs a)
{
    using namespace snc;
    __shared__ uint16_t table[MAX_HASH_TABLE_SIZE];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const uint8_t* __restrict__ in0 = a.srcBase + a.srcOff[block];
    uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
    const int32_t inLen = a.srcLen[block];
    const int32_t outCap = a.dstCap[block];

    int32_t st = 0;
    int32_t output = 0;
    const int64_t bound = 32 + (int64_t)inLen + inLen / 6;  // :69
    if ((int64_t)outCap < bound) {                          // :85-88
        st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNAPPY_MAX_OUTPUT);
    }
    else {
        // writeUncompressedLength :383-411
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
            const uint8_t* __restrict__ in = in0 + blockAddress;  // positions below are relative to the sub-block
            const int32_t blockLimit = (int32_t)((inLen - blockAddress) < BLOCK_SIZE ? (inLen - blockAddress) : BLOCK_SIZE);
            int32_t input = 0;

            // getHashTableSize :348-361
            int32_t tableSize = blockLimit <= 1 ? 0 : (int32_t)((0x80000000u >> __builtin_clz((uint32_t)(blockLimit - 1))) << 1);
            tableSize = tableSize < 256 ? 256 : (tableSize > MAX_HASH_TABLE_SIZE ? MAX_HASH_TABLE_SIZE : tableSize);
            __syncthreads();
            for (int i = lane; i < tableSize; i += 64) {
                table[i] = 0;
            }
            __syncthreads();
            const int32_t shift = 32 - (31 - __builtin_clz((uint32_t)tableSize));

            int32_t nextEmit = input;
            const int32_t fastInputLimit = blockLimit - INPUT_MARGIN_BYTES;
            while (input <= fastInputLimit) {
                int32_t skip = 32;
                int32_t candidate = 0;
                for (input += 1; input + (int32_t)((uint32_t)skip >> 5) <= fastInputLimit; input += (int32_t)((uint32_t)(skip++) >> 5)) {  // :141-159
                    const uint32_t currentInt = ld4(in + input);
                    const int32_t hash = snappy_hash(currentInt, shift);
                    candidate = table[hash];
                    table[hash] = (uint16_t)input;  // every lane stores the same value
                    if (currentInt == ld4(in + candidate)) {
                        break;
                    }
                }
                if (input + (int32_t)((uint32_t)skip >> 5) > fastInputLimit) {
                    break;
                }

                const int32_t literalLength = input - nextEmit;  // :169-175
                output += snappy_literal_header(out, output, literalLength, lane);
                group_copy<64>(out + output, in + nextEmit, literalLength, lane);
                output += literalLength;

                uint32_t inputBytes;
                do {  // :186-219
                    int32_t matched = 4 + wave_count(in, input + 4, candidate + 4, blockLimit, lane);
                    output = snappy_emit_copy(out, output, input - candidate, matched, lane);
                    input += matched;
                    if (input >= fastInputLimit) {
                        break;
                    }
                    const uint64_t longValue = ld8(in + input - 1);
                    const uint32_t prevInt = (uint32_t)longValue;
                    inputBytes = (uint32_t)(longValue >> 8);
                    table[snappy_hash(prevInt, shift)] = (uint16_t)(input - 1);
                    const int32_t curHash = snappy_hash(inputBytes, shift);
                    candidate = table[curHash];
                    table[curHash] = (uint16_t)input;
                } while (inputBytes == ld4(in + candidate));
                nextEmit = input;
            }

            if (nextEmit < blockLimit) {  // :224-229
                const int32_t literalLength = blockLimit - nextEmit;
                output += snappy_literal_header(out, output, literalLength, lane);
                group_copy<64>(out + output, in + nextEmit, literalLength, lane);
                output += literalLength;
            }
        }
    }

    if (lane == 0) {
        a.outLen[block] = st == 0 ? output : 0;
        a.status[block] = st;
        a.errOffset[block] = 0;
    }
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

// Batch-probe variant: 64 steps of the Java search loop per wave step (same scheme as lz4_compress_batch_kernel).
//   after a copy (:199-219): lane 0 = the `input - 1` insert, lane 1 = the re-probe at `input`, lanes 2.. = the probes of
//   the search that follows (:138-162, skip schedule included).
__global__ __launch_bounds__(64) void snappy_compress_batch_kernel(BatchArgs a)
{
    using namespace snc;
    __shared__ uint16_t table[MAX_HASH_TABLE_SIZE];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const uint8_t* __restrict__ in0 = a.srcBase + a.srcOff[block];
    uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
    const int32_t inLen = a.srcLen[block];
    const int32_t outCap = a.dstCap[block];

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
            __syncthreads();
            for (int i = lane; i < tableSize; i += 64) {
                table[i] = 0;
            }
            __syncthreads();
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
                    __syncthreads();

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
    if (lane == 0) {
        a.outLen[block] = st == 0 ? output : 0;
        a.status[block] = st;
        a.errOffset[block] = 0;
    }
}

hipError_t launch_snappy_compress(const BatchArgs& a, hipStream_t stream, int variant)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    if (variant == 0) {
        hipLaunchKernelGGL(snappy_compress_kernel, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a);
    }
    else {
        hipLaunchKernelGGL(snappy_compress_batch_kernel, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace achip
