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

template <typename TableT>
__global__ __launch_bounds__(64) void lz4_compress_kernel(BatchArgs a, int32_t wideOnly)
{
    using namespace lz4c;
    __shared__ TableT table[MAX_TABLE_SIZE];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const int32_t inLen = a.srcLen[block];
    constexpr bool WIDE = sizeof(TableT) == 4;
    // two launches cover a batch: u16 tables for blocks <= 64 KiB, i32 tables for the rest
    if (WIDE ? (inLen <= 65536) : (inLen > 65536)) {
        return;
    }
    (void)wideOnly;
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

hipError_t launch_lz4_compress(const BatchArgs& a, hipStream_t stream, int variant, int maxSrcLenHint)
{
    (void)variant;
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    // maxSrcLenHint: 0 = unknown (launch both table widths), else the largest srcLen in the batch
    if (maxSrcLenHint == 0 || maxSrcLenHint <= 65536) {
        hipLaunchKernelGGL(lz4_compress_kernel<uint16_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, 0);
    }
    if (maxSrcLenHint == 0 || maxSrcLenHint > 65536) {
        hipLaunchKernelGGL(lz4_compress_kernel<int32_t>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, 1);
    }
    return hipGetLastError();
}

}  // namespace achip
