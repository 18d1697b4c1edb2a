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

// ---- sub-blocks side by side (round 6) -----------------------------------------------------------------------------------------------------------
// The Java encoder compresses a buffer in INDEPENDENT sub-blocks of 64 KiB (SnappyRawCompressor.java:93-99: the table is cleared for each, positions count
// from its start, no copy reaches across), and writes them one behind the other.  Until round 5 a buffer was one wavefront's work whatever its length: a
// 4 MB file 0.44 s on a chip that was otherwise idle.  Now every sub-block beyond the first is a work unit of its own:
//   list     (snappy_fan_list_kernel) a thread per buffer: buffers of more than one sub-block are listed, with the first of their extra units
//            (one 64-bit atomic per wavefront carries the count of buffers and the sum of units: list order = unit order, so a unit finds its
//            buffer by a binary search over the listed buffers only);
//   encode   the persistent wavefronts of the two tiers draw UNITS: unit u < nBlocks is buffer u (a listed buffer: its preamble and sub-block 0, in
//            place), unit nBlocks + e is an extra sub-block s >= 1 of a listed buffer, written into the buffer's OWN output at the provisional place
//            preamble + s * FAN_STRIDE behind a 4-byte length (FAN_STRIDE = 65536 + 65536 / 6: the capacity the Java check :85-88 demands -- 32 + n + n / 6
//            -- holds every sub-block at its worst-case distance; a sub-block's output is at most its length + 3);
//   fold     (snappy_fan_fold_kernel) a wavefront per listed buffer moves sub-block 1, 2, ... down behind sub-block 0 (to the left, in order: a
//            piece's destination never reaches the next piece's source) and stores the stream's length.
// The bytes are the serial encoder's by construction.  No host round trip: the unit count lives on the device, the encode grid is persistent.
namespace snfan {
constexpr int32_t FAN_STRIDE = snc::BLOCK_SIZE + snc::BLOCK_SIZE / 6;
constexpr unsigned long long UNITS_MASK = (1ull << 40) - 1ull;
struct State {              // at the scratch's start
    int32_t nextUnit;       // the encode kernel's draw counter
    int32_t pad;
    unsigned long long packed;  // listed buffers << 40 | their extra units
};
__device__ __forceinline__ int32_t preamble_bytes(int32_t inLen)
{
    const uint32_t n = (uint32_t)inLen;
    return n < (1u << 7) ? 1 : (n < (1u << 14) ? 2 : (n < (1u << 21) ? 3 : (n < (1u << 28) ? 4 : 5)));
}
}  // namespace snfan

__global__ __launch_bounds__(256) void snappy_fan_list_kernel(BatchArgs a, snfan::State* state, int32_t* bigItem, int32_t* bigFirst)
{
    using namespace snfan;
    const int lane = threadIdx.x & 63;
    const int32_t perGrid = (int32_t)(gridDim.x * blockDim.x);
    for (int32_t i0 = (int32_t)(blockIdx.x * blockDim.x) + (threadIdx.x & ~63); i0 < a.nBlocks; i0 += perGrid) {  // (uniform per wavefront)
        const int32_t i = i0 + lane;
        const int32_t len = i < a.nBlocks ? a.srcLen[i] : 0;
        const int32_t extra = len > snc::BLOCK_SIZE ? (int32_t)(((int64_t)len + snc::BLOCK_SIZE - 1) / snc::BLOCK_SIZE) - 1 : 0;
        const unsigned long long big = __ballot(extra > 0);
        if (big == 0) {
            continue;
        }
        int32_t incl = extra;  // inclusive scan over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t v = __shfl(incl, lane >= d ? lane - d : lane);
            incl += lane >= d ? v : 0;
        }
        const int32_t total = __shfl(incl, 63);
        unsigned long long old = 0;
        if (lane == 0) {
            old = atomicAdd(&state->packed, ((unsigned long long)__popcll(big) << 40) | (unsigned long long)(uint32_t)total);
        }
        old = ((unsigned long long)(uint32_t)__shfl((int32_t)(old >> 32), 0) << 32) | (uint32_t)__shfl((int32_t)(uint32_t)old, 0);
        if (extra > 0) {
            const int32_t j = (int32_t)(old >> 40) + (int32_t)__popcll(big & ((1ull << lane) - 1ull));
            bigItem[j] = i;
            bigFirst[j] = (int32_t)(old & UNITS_MASK) + incl - extra;
        }
    }
}

__global__ __launch_bounds__(64) void snappy_fan_fold_kernel(BatchArgs a, const snfan::State* state, const int32_t* bigItem)
{
    using namespace snfan;
    const int lane = threadIdx.x;
    const int32_t nBig = (int32_t)(state->packed >> 40);
    for (int32_t j = (int32_t)blockIdx.x; j < nBig; j += (int32_t)gridDim.x) {  // (uniform)
        const int32_t item = bigItem[j];
        if (a.status[item] != 0) {
            continue;  // (the capacity check failed: nothing was written)
        }
        const int32_t len = a.srcLen[item];
        uint8_t* out = a.dstBase + a.dstOff[item];
        const int32_t nSub = (int32_t)(((int64_t)len + snc::BLOCK_SIZE - 1) / snc::BLOCK_SIZE);
        const int32_t pre = preamble_bytes(len);
        int32_t pos = a.outLen[item];  // the preamble and sub-block 0
        wave_sync();
        for (int32_t s = 1; s < nSub; s++) {
            const uint8_t* src = out + pre + (int64_t)s * FAN_STRIDE;
            const int32_t n = (int32_t)ld4(src);
            src += 4;
            uint8_t* dst = out + pos;
            // to the left, a KiB at a time: every lane has read its piece before any lane stores (a piece's destination may reach into its own source)
            for (int32_t base = 0; base < n; base += 1024) {  // (uniform)
                const int32_t at = base + lane * 16;
                const bool whole = at + 16 <= n;
                u32x4 v = {0, 0, 0, 0};
                uint8_t tail[16];
                if (whole) {
                    v = ld16(src + at);
                }
                else {
                    for (int32_t k = 0; at + k < n && k < 16; k++) {
                        tail[k] = src[at + k];
                    }
                }
                wave_sync();
                if (whole) {
                    st16(dst + at, v);
                }
                else {
                    for (int32_t k = 0; at + k < n && k < 16; k++) {
                        dst[at + k] = tail[k];
                    }
                }
                wave_sync();
            }
            pos += n;
        }
        if (lane == 0) {
            a.outLen[item] = pos;
        }
        wave_sync();
    }
}

// Two-tier variant (the default).  The 32 KB hash table allows five wavefronts per CU when it sits in LDS, and the encoder is a
// serial chain per buffer, so throughput is (wavefronts in flight) x (chain speed).  A workgroup here is FOUR wavefronts around one
// LDS table: wavefront 0 encodes with the table in LDS, the other three with tables in global memory (a 32 KB slab each, resident
// in the L2 / Infinity Cache) -- slower chains, but fifteen more of them per CU.  Wavefronts are independent and persistent (each
// draws its next unit from a counter), so the faster ones simply take more.
// fan != nullptr (MW only): units are buffers and the extra sub-blocks of the listed buffers (above).
template <bool MW>  // MW: the "many matches per window" form of the encoder (snappy_compress_mw.h)
__global__ __launch_bounds__(256) void snappy_compress_tiers_kernel(BatchArgs a, uint16_t* slabs, int32_t* nextItem, const snfan::State* fan, const int32_t* bigItem,
                                                                    const int32_t* bigFirst)
{
    using namespace snc;
    __shared__ uint16_t ldsTable[MAX_HASH_TABLE_SIZE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint16_t* const slab = slabs + ((size_t)blockIdx.x * 3 + (wave > 0 ? wave - 1 : 0)) * MAX_HASH_TABLE_SIZE;
    int32_t nBig = 0, nExtra = 0;
    if (MW && fan != nullptr) {
        const unsigned long long packed = uni((uint64_t)fan->packed);
        nBig = (int32_t)(packed >> 40);
        nExtra = (int32_t)(packed & snfan::UNITS_MASK);
    }
    for (;;) {
        int32_t unit = 0;
        if (lane == 0) {
            unit = atomicAdd(nextItem, 1);
        }
        unit = __builtin_amdgcn_readfirstlane(unit);
        if (unit >= a.nBlocks + nExtra) {
            return;
        }
        int32_t block = unit, sub = 0;
        if (unit >= a.nBlocks) {  // an extra sub-block: the listed buffer whose units hold this one (uniform)
            const int32_t e = unit - a.nBlocks;
            int32_t lo = 0, hi = nBig - 1;
            while (lo < hi) {
                const int32_t mid = (lo + hi + 1) >> 1;
                if (uni(bigFirst[mid]) <= e) lo = mid;
                else hi = mid - 1;
            }
            block = uni(bigItem[lo]);
            sub = 1 + e - uni(bigFirst[lo]);
        }
        const uint8_t* __restrict__ in0 = a.srcBase + a.srcOff[block];
        uint8_t* __restrict__ out = a.dstBase + a.dstOff[block];
        const int32_t inLen = uni(a.srcLen[block]);
        int32_t st = 0;
        int32_t output = 0;
        if (MW) {
            const bool fanned = nBig > 0 && inLen > BLOCK_SIZE;  // (a listed buffer)
            const int32_t subLimit = fanned ? sub + 1 : 0x7FFF;
            const int32_t at = sub == 0 ? -1 : snfan::preamble_bytes(inLen) + sub * snfan::FAN_STRIDE + 4;
            if (wave == 0) snappy_compress_buffer_mw(ldsTable, in0, inLen, out, a.dstCap[block], lane, st, output, sub, subLimit, at);
            else snappy_compress_buffer_mw(slab, in0, inLen, out, a.dstCap[block], lane, st, output, sub, subLimit, at);
            if (sub > 0) {
                if (st == 0 && lane == 0) {
                    st4(out + at - 4, (uint32_t)(output - at));
                }
                wave_mem_order();
                continue;  // (status and length are the buffer's own unit's and the fold's)
            }
        }
        else if (wave == 0) {
            snappy_compress_buffer(ldsTable, in0, inLen, out, a.dstCap[block], lane, st, output);
        }
        else {
            snappy_compress_buffer(slab, in0, inLen, out, a.dstCap[block], lane, st, output);
        }
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
int g_snappy_tier_workgroups = SNC_TIER_WORKGROUPS;  // (the persistent grid where the unit count is known on the device only; tools/hostemu makes it small)
// wavefronts of a workgroup whose table lies in memory (0 .. 3; `snappy.compress.mem_waves`).  Their tables are what the kernel's HBM traffic is made of -- 431 GB
// a launch on the corpus batch, 100 x the input: 1 280 workgroups x 3 slabs x 32 KiB = 120 MB of tables, 15 MB per XCD against 4 MB of L2.
int g_snappy_mem_waves = 3;
namespace {
constexpr int64_t SNC_SLABS_BYTES = (int64_t)SNC_TIER_WORKGROUPS * 3 * snc::MAX_HASH_TABLE_SIZE * 2;
}
// [4 KiB: the draw counter, the fan-out's state][the memory tier's table slabs][listed buffers: nBlocks x int32][their first units: nBlocks x int32]
int64_t snappy_compress_scratch_bytes(int32_t nBlocks) { return 4096 + SNC_SLABS_BYTES + 2 * (((int64_t)(nBlocks > 0 ? nBlocks : 1) * 4 + 255) & ~(int64_t)255); }

// variant 0: serial probing, 1: batch probing with the table in LDS (a wavefront per buffer), 2: batch probing, two tiers, 4 (default): many matches per
// window, two tiers, the sub-blocks of buffers beyond 64 KiB side by side (fan = false: in turn, as until round 5)
hipError_t launch_snappy_compress(const BatchArgs& a, hipStream_t stream, int variant, void* scratch, bool fan)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    if (variant == 2 || variant == 4) {
        int32_t* counter = (int32_t*)scratch;
        const hipError_t e = hipMemsetAsync(counter, 0, 64, stream);
        if (e != hipSuccess) return e;
        uint16_t* slabs = (uint16_t*)((uint8_t*)scratch + 4096);
        const unsigned perGroup = (unsigned)(1 + g_snappy_mem_waves);
        const unsigned need = (unsigned)((a.nBlocks + perGroup - 1) / perGroup);
        if (variant == 4 && fan) {
            snfan::State* state = (snfan::State*)scratch;
            int32_t* bigItem = (int32_t*)((uint8_t*)scratch + 4096 + SNC_SLABS_BYTES);
            int32_t* bigFirst = (int32_t*)((uint8_t*)bigItem + (((int64_t)a.nBlocks * 4 + 255) & ~(int64_t)255));
            const unsigned listGrid = (unsigned)((a.nBlocks + 255) / 256);
            hipLaunchKernelGGL(snappy_fan_list_kernel, dim3(listGrid < 1024u ? listGrid : 1024u), dim3(256), 0, stream, a, state, bigItem, bigFirst);
            // (the extra units are known on the device only: the persistent grid is launched whole)
            hipLaunchKernelGGL(snappy_compress_tiers_kernel<true>, dim3((unsigned)g_snappy_tier_workgroups), dim3(64 * (1 + g_snappy_mem_waves)), 0, stream, a, slabs, counter, state, bigItem, bigFirst);
            hipLaunchKernelGGL(snappy_fan_fold_kernel, dim3((unsigned)(a.nBlocks < 4096 ? a.nBlocks : 4096)), dim3(64), 0, stream, a, state, bigItem);
            return hipGetLastError();
        }
        const unsigned grid = need < (unsigned)SNC_TIER_WORKGROUPS ? need : (unsigned)SNC_TIER_WORKGROUPS;
        if (variant == 4) hipLaunchKernelGGL(snappy_compress_tiers_kernel<true>, dim3(grid), dim3(64 * perGroup), 0, stream, a, slabs, counter, nullptr, nullptr, nullptr);
        else hipLaunchKernelGGL(snappy_compress_tiers_kernel<false>, dim3(grid), dim3(64 * perGroup), 0, stream, a, slabs, counter, nullptr, nullptr, nullptr);
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
