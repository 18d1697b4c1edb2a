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
#include "snappy_compress_body.h"
#include "snappy_compress_mw.h"

namespace achip {

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
    snappy_compress_buffer(table, in0, inLen, out, outCap, lane, st, output);
    if (lane == 0) {
        a.outLen[block] = st == 0 ? output : 0;
        a.status[block] = st;
        a.errOffset[block] = 0;
    }
}

// Two-tier variant (the default).  The 32 KB hash table allows five wavefronts per CU when it sits in LDS, and the encoder is a
// serial chain per buffer, so throughput is (wavefronts in flight) x (chain speed).  A workgroup here is FOUR wavefronts around one
// LDS table: wavefront 0 encodes with the table in LDS, the other three with tables in global memory (a 32 KB slab each, resident
// in the L2 / Infinity Cache) -- slower chains, but fifteen more of them per CU.  Wavefronts are independent and persistent (each
// draws its next buffer from a counter), so the faster ones simply take more buffers.
template <bool MW>  // MW: the "many matches per window" form of the encoder (snappy_compress_mw.h)
__global__ __launch_bounds__(256) void snappy_compress_tiers_kernel(BatchArgs a, uint16_t* slabs, int32_t* nextItem)
{
    using namespace snc;
    __shared__ uint16_t ldsTable[MAX_HASH_TABLE_SIZE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint16_t* const slab = slabs + ((size_t)blockIdx.x * 3 + (wave > 0 ? wave - 1 : 0)) * MAX_HASH_TABLE_SIZE;
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
        uint16_t* const table = wave == 0 ? ldsTable : slab;
        if (MW) {
            if (wave == 0) snappy_compress_buffer_mw(ldsTable, in0, a.srcLen[block], out, a.dstCap[block], lane, st, output);
            else snappy_compress_buffer_mw(slab, in0, a.srcLen[block], out, a.dstCap[block], lane, st, output);
        }
        else if (wave == 0) {
            snappy_compress_buffer(ldsTable, in0, a.srcLen[block], out, a.dstCap[block], lane, st, output);
        }
        else {
            snappy_compress_buffer(slab, in0, a.srcLen[block], out, a.dstCap[block], lane, st, output);
        }
        (void)table;
        if (lane == 0) {
            a.outLen[block] = st == 0 ? output : 0;
            a.status[block] = st;
            a.errOffset[block] = 0;
        }
        wave_mem_order();
    }
}

namespace {
constexpr int SNC_TIER_WORKGROUPS = 256 * 5;  // five 32 KB LDS tables per CU
}
int64_t snappy_compress_scratch_bytes() { return 4096 + (int64_t)SNC_TIER_WORKGROUPS * 3 * snc::MAX_HASH_TABLE_SIZE * 2; }

// variant 0: serial probing, 1: batch probing with the table in LDS (a wavefront per buffer), 2 (default): batch probing, two tiers
hipError_t launch_snappy_compress(const BatchArgs& a, hipStream_t stream, int variant, void* scratch)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    if (variant == 2 || variant == 4) {
        int32_t* counter = (int32_t*)scratch;
        const hipError_t e = hipMemsetAsync(counter, 0, 64, stream);
        if (e != hipSuccess) return e;
        const unsigned need = (unsigned)((a.nBlocks + 3) / 4);
        const unsigned grid = need < (unsigned)SNC_TIER_WORKGROUPS ? need : (unsigned)SNC_TIER_WORKGROUPS;
        if (variant == 4) hipLaunchKernelGGL(snappy_compress_tiers_kernel<true>, dim3(grid), dim3(256), 0, stream, a, (uint16_t*)((uint8_t*)scratch + 4096), counter);
        else hipLaunchKernelGGL(snappy_compress_tiers_kernel<false>, dim3(grid), dim3(256), 0, stream, a, (uint16_t*)((uint8_t*)scratch + 4096), counter);
        return hipGetLastError();
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
