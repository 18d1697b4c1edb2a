// zstd_compress_body.h -- the Zstd level-3 encoder's device code (everything but the kernels): shared by zstd_compress.hip (the frame
// compressor) and zstd_stream.hip (the stream writer) as SEPARATE translation units, so that the frame compressor's kernels compile to exactly
// the code they were verified as (with the stream writer in the same unit the inliner's choices changed).  See zstd_compress.hip for the
// description and the reference citations.
#pragma once
#include "achip_device.h"
#include "achip_seqexec.h"  // sx::wave_scan_incl, sx::wave_bcast

namespace achip {

namespace zc {
constexpr int MAX_BLOCK_SIZE = 128 * 1024;
constexpr int MAX_SEQUENCES = MAX_BLOCK_SIZE / 4;
constexpr int HASH_TABLE_INTS = 1 << 17;
constexpr int CHAIN_TABLE_INTS = 1 << 16;
constexpr int64_t SLAB_BYTES = (int64_t)4 * HASH_TABLE_INTS + 4 * CHAIN_TABLE_INTS + (MAX_BLOCK_SIZE + 64) + 3 * 4 * MAX_SEQUENCES + 3 * MAX_SEQUENCES + 64;

__constant__ uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint8_t LL_CODE[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20, 20, 21, 21, 21, 21,
                                    22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
__constant__ uint8_t ML_CODE[128] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
                                     32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39,
                                     40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41,
                                     42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};
__constant__ int16_t LL_DEFAULT_NORM[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ int16_t ML_DEFAULT_NORM[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ int16_t OF_DEFAULT_NORM[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__constant__ int32_t REST_TO_BEAT[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
// level-3 rows of CompressionParameters.DEFAULT_COMPRESSION_PARAMETERS: {windowLog, chainLog, hashLog, searchLength}
__constant__ int32_t LEVEL3[4][4] = {{20, 16, 17, 5}, {18, 16, 16, 4}, {17, 15, 16, 5}, {14, 14, 14, 4}};

struct FseCTable {  // FseCompressionTable.java:22-27
    int16_t nextState[512];
    int32_t deltaNumberOfBits[56];
    int32_t deltaFindState[56];
    int32_t log2Size;
};
struct HufCTable {  // HuffmanCompressionTable.java:29-33
    int16_t values[256];
    uint8_t numberOfBits[256];
    int32_t maxSymbol;
    int32_t maxNumberOfBits;
};
struct Shared {
    FseCTable ll, of, ml, wt;  // SequenceEncodingContext + HuffmanTableWriterWorkspace.fseTable
    FseCTable dflt[3];         // 0 = literal lengths, 1 = offsets, 2 = match lengths (SequenceEncoder.java:66-68)
    HufCTable huf[2];
    int32_t counts[256];
    int16_t norm[256];
    int32_t cumulative[258];
    uint8_t spread[512];
    // NodeTable (2 * 256 - 1 nodes)
    int32_t nodeCount[512];
    int16_t nodeParent[512];
    int16_t nodeSymbol[512];
    uint8_t nodeBits[512];
    int16_t entriesPerRank[16];
    int16_t valuesPerRank[16];
    int32_t rankLast[16];
    uint8_t weights[256];
};

struct Ctx {
    const uint8_t* __restrict__ in;
    int32_t inLen;
    uint8_t* out;
    int32_t outCap;
    int lane;
    int batchProbe;
    int32_t failStatus;  // 0 = ok
    const int32_t* pre;  // match-finder results of this item (two-kernel path) or null
    // per-wave slab
    int32_t* hashTable;
    int32_t* chainTable;
    uint8_t* litBuf;
    int32_t* seqOffset;
    int32_t* seqLitLen;
    int32_t* seqMatchLen;
    uint8_t* codeLL;
    uint8_t* codeML;
    uint8_t* codeOF;
    // parameters
    int32_t windowLog, windowSize, blockSize, chainLog, hashLog, searchLength;
    // RepeatedOffsets
    int32_t offset0, offset1, tempOffset0, tempOffset1;
    int32_t windowBaseOffset;
    // SequenceStore
    int32_t literalsLength, sequenceCount, longLengthField, longLengthPosition;
    // HuffmanCompressionContext (indices into Shared::huf)
    int32_t previousTable, temporaryTable, previousCandidate, temporaryCandidate;
};

#define ZC_FAIL(c)                                                                          \
    {                                                                                       \
        (c).failStatus = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_ZSTD_MAX_OUTPUT);  \
        return -1;                                                                          \
    }
#define ZC_CHECK(c, cond) \
    if (!(cond)) ZC_FAIL(c)
#define ZC_PROPAGATE(x) \
    if ((x) < 0) return -1

__device__ __forceinline__ int32_t highest_bit(uint32_t v) { return 31 - __builtin_clz(v); }
__device__ __forceinline__ int32_t min_table_log(int32_t inputSize, int32_t maxSymbolValue)  // Util.minTableLog
{
    const int32_t a = highest_bit((uint32_t)(inputSize - 1)) + 1;
    const int32_t b = highest_bit((uint32_t)maxSymbolValue) + 2;
    return a < b ? a : b;
}
__device__ __forceinline__ void st_le(uint8_t* p, uint64_t v, int n)
{
    for (int i = 0; i < n; i++) {
        p[i] = (uint8_t)(v >> (8 * i));
    }
}

// ---- BitOutputStream.java: registers + byte-exact stores -------------------------------------------------
// The Java flush() stores 8 bytes and advances by bitCount/8; only the advanced bytes are final (the rest is
// overwritten by the next flush / the next section), so storing exactly those bytes yields the same stream.
struct BitOut {
    uint8_t* base;
    int32_t start, limit, current;  // limit = start + size - 8 (:47)
    uint64_t container;
    int32_t bitCount;
};
__device__ __forceinline__ void bo_init(BitOut& s, uint8_t* base, int32_t start, int32_t size)
{
    s.base = base;
    s.start = start;
    s.limit = start + size - 8;
    s.current = start;
    s.container = 0;
    s.bitCount = 0;
}
__device__ __forceinline__ void bo_add(BitOut& s, int32_t value, int32_t bits)  // addBits :52-56
{
    const uint64_t mask = (1ull << bits) - 1ull;  // bits <= 31
    s.container |= ((uint64_t)(int64_t)value & mask) << (s.bitCount & 63);
    s.bitCount += bits;
}
__device__ __forceinline__ void bo_add_fast(BitOut& s, int32_t value, int32_t bits)  // addBitsFast :61-65
{
    s.container |= (uint64_t)(int64_t)value << (s.bitCount & 63);
    s.bitCount += bits;
}
__device__ __forceinline__ void bo_flush(BitOut& s)  // :67-80
{
    const int32_t bytes = (int32_t)((uint32_t)s.bitCount >> 3);
    int32_t n = bytes;
    if (s.current + n > s.limit + 8) {
        n = s.limit + 8 - s.current;  // never store past the stream's buffer
    }
    if (n == 8) {
        st8(s.base + s.current, s.container);
    }
    else if (n > 0) {
        st_le(s.base + s.current, s.container, n);
    }
    s.current += bytes;
    if (s.current > s.limit) {
        s.current = s.limit;
    }
    s.bitCount &= 7;
    s.container >>= ((bytes * 8) & 63);
}
// flush() as the Java code does it -- 8 bytes, of which bitCount / 8 are final -- for a stream whose buffer this wavefront (or lane) owns up
// to `safeEnd`: what lies behind the final bytes is this stream's own and is rewritten by its next flush.  Close to safeEnd: exact bytes.
__device__ __forceinline__ void bo_flush_wide(BitOut& s, int32_t safeEnd)
{
    const int32_t bytes = (int32_t)((uint32_t)s.bitCount >> 3);
    if (s.current + 8 <= safeEnd && s.current <= s.limit) {
        st8(s.base + s.current, s.container);
    }
    else {
        int32_t n = bytes;
        if (s.current + n > s.limit + 8) {
            n = s.limit + 8 - s.current;
        }
        if (n > 0) {
            st_le(s.base + s.current, s.container, n);
        }
    }
    s.current += bytes;
    if (s.current > s.limit) {
        s.current = s.limit;
    }
    s.bitCount &= 7;
    s.container >>= ((bytes * 8) & 63);
}
__device__ __forceinline__ int32_t bo_close(BitOut& s)  // :82-92
{
    bo_add_fast(s, 1, 1);
    bo_flush(s);
    if (s.current >= s.limit) {
        return 0;
    }
    if (s.bitCount > 0) {
        s.base[s.current] = (uint8_t)s.container;  // the partial last byte
    }
    return (s.current - s.start) + (s.bitCount > 0 ? 1 : 0);
}

// ---- FseCompressionTable ----------------------------------------------------------------------------------
__device__ void fse_init_rle(FseCTable& t, int32_t symbol)  // :46-55
{
    t.log2Size = 0;
    t.nextState[0] = 0;
    t.nextState[1] = 0;
    t.deltaFindState[symbol] = 0;
    t.deltaNumberOfBits[symbol] = 0;
}

// FseCompressionTable.initialize :57-117 by the wavefront (norm read through a pointer so the predefined distributions can come from constant
// memory; maxSymbol < 64, table log <= 9).  The Java method spreads the symbols over the table by a walk, then hands every symbol's states its
// slots in position order.  Closed forms, as in the decoder's table build (zstd_dec_common.h fse_build):
//   * per SYMBOL (a lane each): slots before it = states before it (one for a "less than one" symbol, which takes a top position in symbol order),
//     the two deltas from that running total;
//   * per POSITION (a lane each, 64 at a time): the walk  position = (position + step) & mask  makes position u its j(u)-th stop, j(u) = u * step^-1
//     mod size; stops on the top positions are skipped, so u is the k-th position filled, k = j(u) - (top positions stopped at earlier); the symbol
//     is the one whose run of filled positions covers k;
//   * nextState[slot of the symbol + rank of u among the symbol's positions] = size + u, the rank by ballots over 64 positions and a running count.
__device__ void fse_initialize(Shared& sh, FseCTable& t, const int16_t* norm, int32_t maxSymbol, int32_t tableLog)
{
    const int lane = (int)threadIdx.x & 63;
    const int32_t tableSize = 1 << tableLog, mask = tableSize - 1;
    int32_t* const filledBefore = &sh.cumulative[0];   // [s]: positions the walk fills before symbol s's
    int32_t* const slotOf = &sh.cumulative[64];        // [s]: the symbol's first slot, then its next free one
    int32_t* const seen = &sh.cumulative[128];         // [s]: the symbol's positions below the 64 under way
    wave_sync();
    // ---- per symbol ----
    const bool mine = lane <= maxSymbol;
    const int32_t n = mine ? (int32_t)norm[lane] : 0;
    const bool isLow = n == -1;
    const int32_t states = isLow ? 1 : (n > 0 ? n : 0);
    const int32_t walked = n > 0 ? n : 0;
    const int32_t slot = sx::wave_scan_incl(states, lane) - states;
    const int32_t filled = sx::wave_scan_incl(walked, lane) - walked;
    const unsigned long long lowMask = __ballot(isLow);
    const int32_t lows = (int32_t)__popcll(lowMask);
    const int32_t high = tableSize - 1 - lows;  // the last position the walk may fill
    if (mine) {
        filledBefore[lane] = filled;
        slotOf[lane] = slot;
        seen[lane] = 0;
        if (isLow) {
            sh.spread[tableSize - 1 - (int32_t)__popcll(lowMask & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        }
        if (n == 0) {
            t.deltaNumberOfBits[lane] = ((tableLog + 1) << 16) - tableSize;
        }
        else if (states == 1) {
            t.deltaNumberOfBits[lane] = (tableLog << 16) - tableSize;
            t.deltaFindState[lane] = slot - 1;
        }
        else {
            const int32_t maxBitsOut = tableLog - highest_bit((uint32_t)(n - 1));
            t.deltaNumberOfBits[lane] = (maxBitsOut << 16) - (n << maxBitsOut);
            t.deltaFindState[lane] = slot - n;
        }
    }
    if (lane == 0) {
        t.log2Size = tableLog;
    }
    wave_sync();
    // ---- per position: its symbol ----
    const uint32_t step = (uint32_t)((tableSize >> 1) + (tableSize >> 3) + 3);
    uint32_t inv = step;  // step^-1 mod 2^32 by Newton's iteration (step is odd: step * step = 1 mod 8)
    inv *= 2u - step * inv;
    inv *= 2u - step * inv;
    inv *= 2u - step * inv;
    inv *= 2u - step * inv;
    for (int32_t u = lane; u <= high; u += 64) {
        const int32_t j = (int32_t)(((uint32_t)u * inv) & (uint32_t)mask);
        int32_t k = j;
        for (int32_t v = high + 1; v < tableSize; v++) {  // (uniform bounds)
            k -= (int32_t)(((uint32_t)v * inv) & (uint32_t)mask) < j ? 1 : 0;
        }
        int32_t lo = 0, hi = maxSymbol;  // the last symbol whose filled-before count is <= k (symbols without positions share their successor's: the last wins)
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (filledBefore[mid] <= k) lo = mid;
            else hi = mid - 1;
        }
        sh.spread[u] = (uint8_t)lo;
    }
    wave_sync();
    // ---- per position: its slot ----
    for (int32_t base = 0; base < tableSize; base += 64) {  // (uniform)
        const int32_t u = base + lane;
        const bool valid = u < tableSize;
        const uint32_t symbol = valid ? (uint32_t)sh.spread[u] : (0x100u + (uint32_t)lane);  // (lanes without a position match nobody)
        unsigned long long same = ~0ull;
#pragma unroll
        for (int bit = 0; bit < 9; bit++) {
            const bool one = ((symbol >> bit) & 1u) != 0;
            const unsigned long long m = __ballot(one);
            same &= one ? m : ~m;
        }
        const int32_t before = (int32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (valid) {
            t.nextState[slotOf[symbol] + seen[symbol] + before] = (int16_t)(tableSize + u);
        }
        wave_sync();  // (every lane has read the running counts)
        if (valid && before == 0) {
            seen[symbol] += (int32_t)__popcll(same);
        }
        wave_sync();
    }
}
__device__ __forceinline__ int32_t fse_begin(const FseCTable& t, int32_t symbol)  // :119-124
{
    const int32_t d = t.deltaNumberOfBits[symbol];
    const int32_t outputBits = (int32_t)((uint32_t)(d + (1 << 15)) >> 16);
    const int32_t base = (int32_t)((uint32_t)((outputBits << 16) - d) >> (outputBits & 31));
    return t.nextState[base + t.deltaFindState[symbol]];
}
__device__ __forceinline__ int32_t fse_encode(const FseCTable& t, BitOut& s, int32_t state, int32_t symbol)  // :126-131
{
    const int32_t outputBits = (int32_t)((uint32_t)(state + t.deltaNumberOfBits[symbol]) >> 16);
    bo_add(s, state, outputBits);
    return t.nextState[(int32_t)((uint32_t)state >> (outputBits & 31)) + t.deltaFindState[symbol]];
}
__device__ __forceinline__ void fse_finish(const FseCTable& t, BitOut& s, int32_t state)  // :133-137
{
    bo_add(s, state, t.log2Size);
    bo_flush(s);
}

// ---- FiniteStateEntropy (compression side) ------------------------------------------------------------------
__device__ int32_t fse_optimal_table_log(int32_t maxTableLog, int32_t inputSize, int32_t maxSymbol)  // :236-255
{
    int32_t result = maxTableLog;
    const int32_t a = highest_bit((uint32_t)(inputSize - 1)) - 2;
    result = a < result ? a : result;
    const int32_t b = min_table_log(inputSize, maxSymbol);
    result = b > result ? b : result;
    result = result < 5 ? 5 : result;
    result = result > 12 ? 12 : result;
    return result;
}

// normalizeCounts2 :318-405 (the rarely taken second method) with a lane per symbol (maxSymbol < 64).  What the Java loops carry along are
// sums over the symbols -- `distributed` (symbols given 1 / -1), `total` (what is left for the others) -- and, in the last loop, a running sum
// of counts x rStep over the unassigned symbols: wave sums and one 64-bit wave scan.  The two corner endings: every symbol is a low one (the
// first largest count takes the rest), and total == 0 (the rest goes round-robin over the positive symbols: each gets rest / K, the first
// rest % K of them one more).  Every lane calls it (wave-uniform); the result is in norm[0 .. maxSymbol] after the caller's wave_sync().
__device__ __forceinline__ int64_t wave_scan_incl_i64(int64_t x, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int32_t)(uint32_t)x, d), hi = (uint32_t)__shfl_up((int32_t)((uint64_t)x >> 32), d);
        x += lane >= d ? (int64_t)(((uint64_t)hi << 32) | lo) : 0;
    }
    return x;
}
__device__ __forceinline__ int32_t wave_sum(int32_t x, int lane) { return sx::wave_bcast(sx::wave_scan_incl(x, lane), 63); }

__device__ void fse_normalize_counts2(int16_t* norm, int32_t tableLog, const int32_t* counts, int32_t total, int32_t maxSymbol, int lane)
{
    const bool mine = lane <= maxSymbol;
    const int32_t cnt = mine ? counts[lane] : 0;
    const int32_t lowThreshold = (int32_t)((uint32_t)total >> tableLog);
    int32_t lowOne = (int32_t)((uint32_t)(total * 3) >> (tableLog + 1));
    // first pass: 0 stays 0, counts up to lowThreshold get -1, up to lowOne get 1, the rest waits
    int32_t value = 0;
    bool open = false;  // not yet assigned
    if (cnt != 0) {
        if (cnt <= lowThreshold) value = -1;
        else if (cnt <= lowOne) value = 1;
        else open = true;
    }
    bool low = mine && !open;  // counted in `distributed` (zeros are not: they add nothing to either sum)
    int32_t distributed = wave_sum(mine && cnt != 0 && !open ? 1 : 0, lane);
    total -= wave_sum(mine && cnt != 0 && !open ? cnt : 0, lane);
    const int32_t normalizationFactor = 1 << tableLog;
    int32_t toDistribute = normalizationFactor - distributed;
    if ((total / toDistribute) > lowOne) {  // (uniform) :351-361 -- a second, wider round of ones
        lowOne = (total * 3) / (toDistribute * 2);
        const bool now = open && cnt <= lowOne;
        if (now) {
            value = 1;
            open = false;
        }
        distributed += wave_sum(now ? 1 : 0, lane);
        total -= wave_sum(now ? cnt : 0, lane);
        toDistribute = normalizationFactor - distributed;
    }
    (void)low;
    if (distributed == maxSymbol + 1) {  // (uniform) :363-378 -- nobody is left: the first of the largest counts takes the rest
        int32_t top = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t other = __shfl_xor(top, d);
            top = other > top ? other : top;
        }
        const int32_t first = top > 0 ? (int32_t)__builtin_ctzll(__ballot(mine && cnt == top)) : 0;
        if (lane == first) {
            value += toDistribute;
        }
    }
    else if (total == 0) {  // (uniform) :380-389 -- round-robin over the positive symbols, starting at symbol 0
        const bool positive = mine && value > 0;
        const unsigned long long pm = __ballot(positive);
        const int32_t k = (int32_t)__popcll(pm);
        if (positive && k > 0) {
            const int32_t rank = (int32_t)__popcll(pm & ((1ull << lane) - 1ull));
            value += toDistribute / k + (rank < toDistribute % k ? 1 : 0);
        }
    }
    else {  // :391-404 -- the rest in proportion, by a running total of count x rStep in 62 - tableLog fixed-point bits
        const int64_t vStepLog = 62 - tableLog;
        const int64_t mid = (1LL << (vStepLog - 1)) - 1;
        const int64_t rStep = (((1LL << vStepLog) * toDistribute) + mid) / total;
        const int64_t weight = open ? (int64_t)cnt * rStep : 0;
        const int64_t end = mid + wave_scan_incl_i64(weight, lane);
        if (open) {
            value = (int32_t)((uint64_t)end >> vStepLog) - (int32_t)((uint64_t)(end - weight) >> vStepLog);
        }
    }
    if (mine) {
        norm[lane] = (int16_t)value;
    }
}

// normalizeCounts :257-316 with a lane per symbol (every caller has at most 53 symbols: sequence codes, Huffman weights).  The Java loop's
// running values are a sum and a first-maximum over the symbols: `stillToDistribute` = table size - (probabilities + one per low symbol),
// `largest` = the first symbol whose probability exceeds all before it (:291-294; symbol 0 when none is rated).  The second method
// (:318-405: when the correction would take more than half of the largest probability) is rare; it is lane-parallel as well (above).
__device__ void fse_normalize_counts(int16_t* norm, int32_t tableLog, const int32_t* counts, int32_t total, int32_t maxSymbol, int lane)
{
    const int64_t scale = 62 - tableLog;
    const int64_t step = (1LL << 62) / total;
    const int64_t vstep = 1LL << (scale - 20);
    const int32_t lowThreshold = (int32_t)((uint32_t)total >> tableLog);
    const bool mine = lane <= maxSymbol;  // (maxSymbol < 64)
    const int32_t cnt = mine ? counts[lane] : 0;
    int32_t value = 0, taken = 0, rated = 0;
    if (cnt != 0) {
        if (cnt <= lowThreshold) {
            value = -1;
            taken = 1;
        }
        else {
            int16_t probability = (int16_t)((uint64_t)((int64_t)cnt * step) >> scale);
            if (probability < 8) {
                const int64_t restToBeat = vstep * REST_TO_BEAT[probability];
                const int64_t delta = (int64_t)cnt * step - (((int64_t)probability) << scale);
                if (delta > restToBeat) {
                    probability++;
                }
            }
            value = probability;
            taken = probability;
            rated = probability;
        }
    }
    const int32_t stillToDistribute = (1 << tableLog) - sx::wave_bcast(sx::wave_scan_incl(taken, lane), 63);
    int32_t top = rated;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t other = __shfl_xor(top, d);
        top = other > top ? other : top;
    }
    const unsigned long long holders = __ballot(rated == top);
    const int32_t largest = top > 0 ? (int32_t)__builtin_ctzll(holders) : 0;
    const int32_t largestValue = __shfl(value, largest);
    if (-stillToDistribute >= (int32_t)((uint32_t)largestValue >> 1)) {  // (uniform)
        fse_normalize_counts2(norm, tableLog, counts, total, maxSymbol, lane);
    }
    else if (mine) {
        norm[lane] = (int16_t)(lane == largest ? value + stillToDistribute : value);
    }
    wave_sync();
}

// writeNormalizedCounts :407-521 with a lane per symbol (maxSymbol < 64); returns bytes written or -1.
// The Java loop appends bit fields to a 32-bit accumulator and stores two bytes whenever more than 16 bits are pending -- the stream is the
// plain concatenation of the fields, least significant bit first, and everything a field depends on is a prefix sum over the symbols before it:
//   remaining(s) = tableSize + 1 - sum of |norm[i]| for i < s        (the loop runs while remaining > 1)
//   threshold(s) = the largest power of two <= remaining(s), at most tableSize; a field is log2(threshold) + 1 bits, one less for a small value
//   a symbol is written unless it is a zero BEHIND a zero: a run of zeros is its first zero's field followed by the run's length -- 16 one-bits
//   per 24 further zeros, two one-bits per 3, two bits for the rest (:437-466)
// so every lane builds its symbol's bits (<= 62: a field of <= 10 bits, a run code of <= 48, the 4-bit table log in front of symbol 0), a scan
// of the widths places them, ds_or puts them into a zeroed window of words (sh.cumulative: idle between a table's initialisation and the next),
// whole bytes leave with one store per lane.  The size checks of the Java loop (two bytes must fit at every store, and once more at the end)
// amount to 2 * floor((bits - 1) / 16) + 2 <= outputSize.
__device__ int32_t fse_write_normalized_counts(Ctx& c, Shared& sh, uint8_t* base, int32_t outputAddress, int32_t outputSize, const int16_t* norm, int32_t maxSymbol, int32_t tableLog)
{
    const int lane = c.lane;
    uint32_t* const window = (uint32_t*)&sh.cumulative[0];  // 64 words
    wave_sync();
    window[lane] = 0;
    const bool mine = lane <= maxSymbol;
    const int32_t value = mine ? (int32_t)norm[lane] : 0;
    const int32_t magnitude = value < 0 ? -value : value;
    const int32_t tableSize = 1 << tableLog;
    const int32_t before = sx::wave_scan_incl(magnitude, lane) - magnitude;
    const int32_t remaining = tableSize + 1 - before;
    const bool inLoop = mine && remaining > 1;
    const unsigned long long zeros = __ballot(mine && value == 0);
    const unsigned long long nonZeros = __ballot(mine && value != 0);
    const bool behindZero = lane > 0 && ((zeros >> (lane - 1)) & 1ull) != 0;
    const bool written = inLoop && !(value == 0 && behindZero);
    uint64_t bits = 0;
    int32_t width = 0;
    if (written) {
        int32_t threshold = 1 << highest_bit((uint32_t)remaining);
        threshold = threshold > tableSize ? tableSize : threshold;
        const int32_t fieldBits = highest_bit((uint32_t)threshold) + 1;
        const int32_t max = (2 * threshold - 1) - remaining;
        int32_t count = value + 1;
        if (count >= threshold) {
            count += max;
        }
        bits = (uint64_t)(uint32_t)count;
        width = fieldBits - (count < max ? 1 : 0);
        if (value == 0) {
            // the zeros that follow this one, up to the next symbol that is not zero (there is one: remaining > 1)
            const unsigned long long above = nonZeros >> lane;  // (bit 0 is this lane: a zero)
            const int32_t run = above != 0 ? (int32_t)__builtin_ctzll(above) - 1 : 0;
            const int32_t ones = 16 * (run / 24) + 2 * ((run % 24) / 3);
            const uint64_t code = ((1ull << ones) - 1ull) | ((uint64_t)((run % 24) % 3) << ones);
            bits |= code << width;
            width += ones + 2;
        }
    }
    if (lane == 0) {  // the table log leads the stream (:414-416)
        bits = (bits << 4) | (uint64_t)(uint32_t)(tableLog - 5);
        width += 4;
    }
    const int32_t end = sx::wave_scan_incl(width, lane);
    const int32_t total = sx::wave_bcast(end, 63);
    // the loop must have ended inside the table (:518-520: "Error"; normalized counts that do not add up to the table size would run off its end)
    const int32_t last = 63 - (int32_t)__builtin_clzll(__ballot(inLoop) | 1ull);
    const int32_t lastMagnitude = sx::wave_bcast(magnitude, last);
    const int32_t lastRemaining = sx::wave_bcast(remaining, last);
    ZC_CHECK(c, lastRemaining - lastMagnitude <= 1);
    ZC_CHECK(c, 2 * ((total - 1) / 16) + 2 <= outputSize);
    wave_sync();
    if (width > 0) {
        const int32_t at = end - width;
        const int32_t word = at >> 5, shift = at & 31;
        atomicOr(&window[word], (uint32_t)(bits << shift));
        const uint64_t rest = shift == 0 ? (bits >> 32) : (bits >> (32 - shift));
        if ((uint32_t)rest != 0) {
            atomicOr(&window[word + 1], (uint32_t)rest);
        }
        if ((uint32_t)(rest >> 32) != 0) {
            atomicOr(&window[word + 2], (uint32_t)(rest >> 32));
        }
    }
    wave_sync();
    const int32_t bytes = (total + 7) / 8;
    for (int32_t k = lane; k < bytes; k += 64) {
        base[outputAddress + k] = (uint8_t)(window[k >> 2] >> (8 * (k & 3)));
    }
    wave_sync();
    return bytes;
}

// FiniteStateEntropy.compress :158-234 over the Huffman weights (<= 255 symbols, LDS)
__device__ int32_t fse_compress_weights(Ctx& c, uint8_t* base, int32_t outputAddress, int32_t outputSize, const uint8_t* in, int32_t inputSize, const FseCTable& table)
{
    ZC_CHECK(c, outputSize >= 8);
    int32_t input = inputSize;
    if (inputSize <= 2) {
        return 0;
    }
    BitOut s;
    bo_init(s, base, outputAddress, outputSize);
    int32_t state1, state2;
    if ((inputSize & 1) != 0) {
        input--;
        state1 = fse_begin(table, in[input]);
        input--;
        state2 = fse_begin(table, in[input]);
        input--;
        state1 = fse_encode(table, s, state1, in[input]);
        bo_flush(s);
    }
    else {
        input--;
        state2 = fse_begin(table, in[input]);
        input--;
        state1 = fse_begin(table, in[input]);
    }
    inputSize -= 2;
    if ((inputSize & 2) != 0) {
        input--;
        state2 = fse_encode(table, s, state2, in[input]);
        input--;
        state1 = fse_encode(table, s, state1, in[input]);
        bo_flush(s);
    }
    while (input > 0) {
        input--;
        state2 = fse_encode(table, s, state2, in[input]);
        input--;
        state1 = fse_encode(table, s, state1, in[input]);
        input--;
        state2 = fse_encode(table, s, state2, in[input]);
        input--;
        state1 = fse_encode(table, s, state1, in[input]);
        bo_flush(s);
    }
    fse_finish(table, s, state2);
    fse_finish(table, s, state1);
    return bo_close(s);
}

// ---- histograms: lanes split the input, LDS atomics (Histogram.java:21-65) -----------------------------------
__device__ void wave_histogram(Shared& sh, const uint8_t* in, int32_t n, int32_t bins, int lane)
{
    __syncthreads();
    for (int32_t i = lane; i < bins; i += 64) {
        sh.counts[i] = 0;
    }
    __syncthreads();
    for (int32_t i = lane; i < n; i += 64) {
        atomicAdd(&sh.counts[in[i]], 1);
    }
    __syncthreads();
}
__device__ __forceinline__ int32_t find_max_symbol(const Shared& sh, int32_t maxSymbol)
{
    while (sh.counts[maxSymbol] == 0) {
        maxSymbol--;
    }
    return maxSymbol;
}
__device__ __forceinline__ int32_t find_largest_count(const Shared& sh, int32_t maxSymbol)
{
    int32_t max = 0;
    for (int32_t i = 0; i <= maxSymbol; i++) {
        const int32_t v = sh.counts[i];
        max = v > max ? v : max;
    }
    return max;
}

// ---- Huffman compression table -------------------------------------------------------------------------------
__device__ int32_t huf_optimal_number_of_bits(int32_t maxNumberOfBits, int32_t inputSize, int32_t maxSymbol)  // :42-57
{
    int32_t result = maxNumberOfBits;
    const int32_t a = highest_bit((uint32_t)(inputSize - 1)) - 1;
    result = a < result ? a : result;
    const int32_t b = min_table_log(inputSize, maxSymbol);
    result = b > result ? b : result;
    result = result < 5 ? 5 : result;
    result = result > 12 ? 12 : result;
    return result;
}

// buildTree :105-190 (counts in sh.counts; returns lastNonZero).  What the Java method computes, in three parts:
//   order   the symbols by falling count, equal counts by rising symbol -- except that symbol 0 keeps position 0 whatever its count (the
//           insertion sort never moves an entry below position 1: `position > 1`).  Here every lane RANKS its symbols: position = 1 + symbols
//           (other than 0) with a larger count + earlier symbols (other than 0) with the same count; no entry is ever moved.
//   merge   Huffman's two queues: the leaves from the rarest up (position lastNonZero down to 0) and the inner nodes in the order they were made
//           (from 256 up), the smaller head first, an inner node on a tie.  A chain by construction: one lane walks it.
//   depth   a node's code length is its distance from the root: pointer jumping over the parent links (every lane eight nodes: depth += depth of
//           where the link points, link = the link's link, until every link points at the root) instead of a walk down from the root.
__device__ int32_t huf_build_tree(Shared& sh, int32_t maxSymbol)
{
    const int lane = (int)threadIdx.x & 63;
    constexpr int32_t INNER = 256;  // the first inner node
    __syncthreads();
    for (int32_t i = lane; i < 512; i += 64) {  // NodeTable.reset()
        sh.nodeCount[i] = 0;
        sh.nodeParent[i] = 0;
        sh.nodeSymbol[i] = 0;
        sh.nodeBits[i] = 0;
    }
    __syncthreads();
    // ---- order ----
    int32_t mine[4], place[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int32_t sym = lane + 64 * j;
        mine[j] = sym <= maxSymbol ? sh.counts[sym] : -1;
        place[j] = sym == 0 ? 0 : 1;
    }
    for (int32_t t = 1; t <= maxSymbol; t++) {
        const int32_t other = sh.counts[t];  // (one address for the wavefront: a broadcast)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int32_t sym = lane + 64 * j;
            place[j] += (other > mine[j] || (other == mine[j] && t < sym)) ? 1 : 0;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int32_t sym = lane + 64 * j;
        if (sym <= maxSymbol) {
            const int32_t at = sym == 0 ? 0 : place[j];
            sh.nodeCount[at] = mine[j];
            sh.nodeSymbol[at] = (int16_t)sym;
        }
    }
    __syncthreads();
    // the last position that holds a count
    int32_t lastNonZero = 0;
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        const int32_t pos = lane + 64 * j;
        const unsigned long long holders = __ballot(pos <= maxSymbol && sh.nodeCount[pos] != 0);
        if (holders != 0 && lastNonZero == 0) {
            lastNonZero = 64 * j + 63 - (int32_t)__builtin_clzll(holders);
        }
    }
    const int32_t root = INNER + lastNonZero - 1;
    // ---- merge ----
    if (lane == 0) {
        int32_t leaf = lastNonZero;  // head of the leaf queue (runs down to 0)
        int32_t inner = INNER;       // head of the inner-node queue
        for (int32_t made = INNER; made <= root; made++) {
            int32_t child[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                // (the inner queue is empty while inner == made: its head then counts as larger than any leaf -- the Java table holds 1 << 30 there)
                const bool takeLeaf = leaf >= 0 && (inner >= made || sh.nodeCount[leaf] < sh.nodeCount[inner]);
                child[k] = takeLeaf ? leaf-- : inner++;
            }
            sh.nodeCount[made] = sh.nodeCount[child[0]] + sh.nodeCount[child[1]];
            sh.nodeParent[child[0]] = (int16_t)made;
            sh.nodeParent[child[1]] = (int16_t)made;
        }
    }
    __syncthreads();
    // ---- depth ---- (nodes in use: the leaves 0 .. lastNonZero, the inner nodes 256 .. root; the root links to itself)
    int32_t link[8], depth[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int32_t n = lane + 64 * j;
        const bool used = (n <= lastNonZero) || (n >= INNER && n < root);
        link[j] = used ? (int32_t)sh.nodeParent[n] : root;
        depth[j] = used ? 1 : 0;
    }
    for (;;) {  // (uniform)
        bool moved = false;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int32_t n = lane + 64 * j;
            sh.nodeParent[n] = (int16_t)link[j];
            sh.nodeBits[n] = (uint8_t)depth[j];
            moved |= link[j] != root;
        }
        __syncthreads();
        if (__ballot(moved) == 0) {
            break;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int32_t up = link[j];
            depth[j] += (int32_t)sh.nodeBits[up];
            link[j] = (int32_t)sh.nodeParent[up];
        }
        __syncthreads();
    }
    return lastNonZero;
}

// setMaxHeight :294-390.  Code lengths beyond maxNumberOfBits are cut to it (the leaves are in order of falling count, so these are the last
// ones); every cut overdraws the Kraft budget by baseCost - 2^(largestBits - length), and the debt -- in units of the deepest allowed code -- is paid
// back by making other leaves one bit longer: always the LAST leaf of some shorter length (rankLast[d] for length maxNumberOfBits - d), chosen
// per step by comparing what a lengthening costs in weighted length.  Here: the cuts, their cost and the rankLast table by ballots over the
// leaves (a lane per leaf, four rounds); the repayment holds rankLast[d] in lane d, a step's choice is two ballots (where the downward scan
// stops, the first rank at or above it that has a leaf) and only the chosen lanes change anything.
__device__ int32_t huf_set_max_height(Shared& sh, int32_t lastNonZero, int32_t maxNumberOfBits)
{
    const int lane = (int)threadIdx.x & 63;
    const int32_t largestBits = sh.nodeBits[lastNonZero];
    if (largestBits <= maxNumberOfBits) {  // (uniform)
        return largestBits;
    }
    constexpr int32_t NONE = (int32_t)0xF0F0F0F0;
    // ---- the cuts: the run of leaves longer than allowed at the end of the table, and the first leaf below it that is shorter than allowed ----
    int32_t bitsOf[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int32_t pos = lane + 64 * j;
        bitsOf[j] = pos <= lastNonZero ? (int32_t)sh.nodeBits[pos] : 0;
    }
    auto highest = [&](auto&& pred) -> int32_t {  // the highest leaf position <= lastNonZero for which pred(j) holds on its lane (-1: none)
        int32_t found = -1;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const unsigned long long m = __ballot(lane + 64 * j <= lastNonZero && pred(j));
            if (m != 0 && found < 0) {
                found = 64 * j + 63 - (int32_t)__builtin_clzll(m);
            }
        }
        return found;
    };
    const int32_t keep = highest([&](int j) { return bitsOf[j] <= maxNumberOfBits; });  // everything above it is cut
    const int32_t baseCost = 1 << (largestBits - maxNumberOfBits);
    int32_t cost = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int32_t pos = lane + 64 * j;
        if (pos > keep && pos <= lastNonZero) {
            cost += baseCost - (1 << (largestBits - bitsOf[j]));
            bitsOf[j] = maxNumberOfBits;
            sh.nodeBits[pos] = (uint8_t)maxNumberOfBits;
        }
    }
    int32_t totalCost = (int32_t)((uint32_t)wave_sum(cost, lane) >> (largestBits - maxNumberOfBits));
    int32_t n = highest([&](int j) { return lane + 64 * j <= keep && bitsOf[j] != maxNumberOfBits; });  // the last leaf shorter than allowed
    // ---- rankLast[d]: the last leaf of length maxNumberOfBits - d, where no later leaf (up to n) is as short or shorter ----
    // (lane d works out its own entry: the highest position of its length, valid if every shorter length's highest position lies below it)
    int32_t lastOf = -1;  // lane b: the highest position <= n whose length is b
    for (int32_t b = 1; b < maxNumberOfBits; b++) {  // (uniform; at most 11 rounds of four ballots)
        const int32_t at = highest([&](int j) { return lane + 64 * j <= n && bitsOf[j] == b; });
        lastOf = lane == b ? at : lastOf;
    }
    int32_t shorterAbove = -1;  // the highest position of any shorter length
    for (int32_t b = 1; b < maxNumberOfBits; b++) {
        const int32_t at = __shfl(lastOf, b);
        shorterAbove = (b < maxNumberOfBits - lane && at > shorterAbove) ? at : shorterAbove;
    }
    const int32_t myLength = maxNumberOfBits - lane;  // lane d holds rankLast[d]
    const int32_t ownLast = __shfl(lastOf, myLength > 0 && myLength < 64 ? myLength : 0);
    int32_t rank = (lane >= 1 && myLength >= 1 && ownLast >= 0 && ownLast > shorterAbove) ? ownLast : NONE;
    __syncthreads();
    // ---- repayment ----
    while (totalCost > 0) {  // (uniform)
        const int32_t start = highest_bit((uint32_t)totalCost) + 1;
        const int32_t below = __shfl(rank, lane > 0 ? lane - 1 : 0);  // rankLast[d - 1]
        // the downward scan (:334-349) stops at the first d <= start that has a leaf and whose neighbour below has none or costs at least half as much
        bool stops = false;
        if (lane > 1 && lane <= start && rank != NONE) {
            stops = below == NONE || sh.nodeCount[rank] <= 2 * sh.nodeCount[below];
        }
        const unsigned long long stopMask = __ballot(stops);
        int32_t dec = stopMask != 0 ? 63 - (int32_t)__builtin_clzll(stopMask) : 1;
        // ... and moves up to the first rank that has a leaf (:351-353), 13 when none has
        const unsigned long long have = __ballot(rank != NONE && lane >= dec && lane <= 12);
        dec = have != 0 ? (int32_t)__builtin_ctzll(have) : 13;
        totalCost -= 1 << (dec - 1);
        const int32_t at = __shfl(rank, dec);
        if (at == NONE) {  // (no leaf left to lengthen: the Java loop would index its table with the marker and throw; not reachable from counts that sum up)
            break;
        }
        const int32_t belowDec = __shfl(rank, dec - 1);
        if (lane == dec - 1 && belowDec == NONE) {
            rank = at;  // the lengthened leaf becomes the last of its new length
        }
        if (lane == dec) {
            sh.nodeBits[at] = (uint8_t)(sh.nodeBits[at] + 1);
            rank = (at == 0 || (int32_t)sh.nodeBits[at - 1] != maxNumberOfBits - dec) ? NONE : at - 1;
        }
        __syncthreads();
    }
    // ---- overpaid (:373-388): give a bit back, to the first leaf of the deepest length but one ----
    if (totalCost < 0) {  // (uniform)
        int32_t rank1 = __shfl(rank, 1);
        if (lane == 0) {
            while (totalCost < 0) {
                if (rank1 == NONE) {
                    while ((int32_t)sh.nodeBits[n] == maxNumberOfBits) {
                        n--;
                    }
                    sh.nodeBits[n + 1] = (uint8_t)(sh.nodeBits[n + 1] - 1);
                    rank1 = n + 1;
                }
                else {
                    rank1++;
                    sh.nodeBits[rank1] = (uint8_t)(sh.nodeBits[rank1] - 1);
                }
                totalCost++;
            }
        }
        __syncthreads();
    }
    return maxNumberOfBits;
}

__device__ void huf_table_initialize(Shared& sh, HufCTable& t, int32_t maxSymbol, int32_t maxNumberOfBits)  // initialize :60-103
{
    for (int i = 0; i < 13; i++) {  // workspace.reset()
        sh.entriesPerRank[i] = 0;
        sh.valuesPerRank[i] = 0;
    }
    const int32_t lastNonZero = huf_build_tree(sh, maxSymbol);
    maxNumberOfBits = huf_set_max_height(sh, lastNonZero, maxNumberOfBits);
    for (int32_t node = 0; node < maxSymbol + 1; node++) {
        t.numberOfBits[sh.nodeSymbol[node]] = sh.nodeBits[node];
    }
    for (int32_t n = 0; n <= lastNonZero; n++) {
        const int32_t r = sh.nodeBits[n];
        sh.entriesPerRank[r] = (int16_t)(sh.entriesPerRank[r] + 1);
    }
    int16_t startingValue = 0;
    for (int32_t rank = maxNumberOfBits; rank > 0; rank--) {
        sh.valuesPerRank[rank] = startingValue;
        startingValue = (int16_t)(startingValue + sh.entriesPerRank[rank]);
        startingValue = (int16_t)((uint32_t)(int32_t)startingValue >> 1);
    }
    for (int32_t n = 0; n <= maxSymbol; n++) {
        const int32_t r = t.numberOfBits[n];
        const int16_t v = sh.valuesPerRank[r];
        t.values[n] = v;
        sh.valuesPerRank[r] = (int16_t)(v + 1);
    }
    t.maxSymbol = maxSymbol;
    t.maxNumberOfBits = maxNumberOfBits;
}

// compressWeights :392-436 ; returns size, or -1 on failure
__device__ int32_t huf_compress_weights(Ctx& c, Shared& sh, uint8_t* base, int32_t outputAddress, int32_t outputSize, int32_t weightsLength)
{
    if (weightsLength <= 1) {
        return 0;
    }
    // histogram of <= 255 weights: serial, it is tiny (Histogram.count over workspace.counts[13])
    int32_t wcounts[13];
#pragma unroll
    for (int i = 0; i < 13; i++) wcounts[i] = 0;
    for (int32_t i = 0; i < weightsLength; i++) {
        const int32_t w = sh.weights[i];
#pragma unroll
        for (int k = 0; k < 13; k++) wcounts[k] += (w == k);
    }
    int32_t maxSymbol = 12;
    int32_t maxCount = 0;
    {
        bool found = false;
#pragma unroll
        for (int k = 12; k >= 0; k--) {
            if (!found && wcounts[k] != 0) {
                maxSymbol = k;
                found = true;
            }
        }
#pragma unroll
        for (int k = 0; k < 13; k++) {
            if (k <= maxSymbol) maxCount = wcounts[k] > maxCount ? wcounts[k] : maxCount;
        }
    }
    if (maxCount == weightsLength) {
        return 1;
    }
    if (maxCount == 1) {
        return 0;
    }
    // the generic FSE routines index their inputs dynamically: park the 13 counts in LDS (rankLast is free once
    // setMaxHeight is done; the literal histogram in sh.counts must survive for estimateCompressedSize)
    int32_t* countsLds = &sh.rankLast[0];
#pragma unroll
    for (int k = 0; k < 13; k++) countsLds[k] = wcounts[k];
    int16_t* normLds = &sh.norm[128];  // 13 shorts, disjoint from the sequence normalizedCounts use (index < 53)
    const int32_t tableLog = fse_optimal_table_log(6, weightsLength, maxSymbol);
    fse_normalize_counts(normLds, tableLog, countsLds, weightsLength, maxSymbol, c.lane);
    int32_t output = outputAddress;
    const int32_t outputLimit = outputAddress + outputSize;
    const int32_t headerSize = fse_write_normalized_counts(c, sh, base, output, outputSize, normLds, maxSymbol, tableLog);
    ZC_PROPAGATE(headerSize);
    output += headerSize;
    fse_initialize(sh, sh.wt, normLds, maxSymbol, tableLog);
    const int32_t compressedSize = fse_compress_weights(c, base, output, outputLimit - output, sh.weights, weightsLength, sh.wt);
    ZC_PROPAGATE(compressedSize);
    if (compressedSize == 0) {
        return 0;
    }
    output += compressedSize;
    return output - outputAddress;
}

// HuffmanCompressionTable.write :206-268 ; returns size or -1
__device__ int32_t huf_table_write(Ctx& c, Shared& sh, const HufCTable& t, uint8_t* base, int32_t outputAddress, int32_t outputSize)
{
    int32_t output = outputAddress;
    const int32_t maxNumberOfBits = t.maxNumberOfBits;
    const int32_t maxSymbol = t.maxSymbol;
    for (int32_t symbol = 0; symbol < maxSymbol; symbol++) {
        const int32_t bits = t.numberOfBits[symbol];
        sh.weights[symbol] = bits == 0 ? (uint8_t)0 : (uint8_t)(maxNumberOfBits + 1 - bits);
    }
    int32_t size = huf_compress_weights(c, sh, base, output + 1, outputSize - 1, maxSymbol);
    ZC_PROPAGATE(size);
    if (maxSymbol > 127 && size > 127) {  // Java: AssertionError
        c.failStatus = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED);
        return -1;
    }
    if (size != 0 && size != 1 && size < maxSymbol / 2) {
        base[output] = (uint8_t)size;
        return size + 1;
    }
    const int32_t entryCount = maxSymbol;
    size = (entryCount + 1) / 2;
    ZC_CHECK(c, size + 1 <= outputSize);
    base[output] = (uint8_t)(127 + entryCount);
    output++;
    if (maxSymbol >= 255) {  // Java: weights[255] ArrayIndexOutOfBoundsException
        c.failStatus = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED);
        return -1;
    }
    sh.weights[maxSymbol] = 0;
    for (int32_t i = 0; i < entryCount; i += 2) {
        base[output] = (uint8_t)((sh.weights[i] << 4) + sh.weights[i + 1]);
        output++;
    }
    return output - outputAddress;
}

__device__ bool huf_table_is_valid(const Shared& sh, const HufCTable& t, int32_t maxSymbol)  // :273-286
{
    if (maxSymbol > t.maxSymbol) {
        return false;
    }
    for (int32_t symbol = 0; symbol <= maxSymbol; ++symbol) {
        if (sh.counts[symbol] != 0 && t.numberOfBits[symbol] == 0) {
            return false;
        }
    }
    return true;
}
__device__ int32_t huf_estimate_compressed_size(const Shared& sh, const HufCTable& t, int32_t maxSymbol)  // :288-296
{
    int32_t bits = 0;
    const int32_t lim = maxSymbol < t.maxSymbol ? maxSymbol : t.maxSymbol;
    for (int32_t symbol = 0; symbol <= lim; symbol++) {
        bits += t.numberOfBits[symbol] * sh.counts[symbol];
    }
    return (int32_t)((uint32_t)bits >> 3);
}

// total code length of in[0..n) under table t: lanes split the input, wave reduction
__device__ int32_t wave_code_bits(const HufCTable& t, const uint8_t* in, int32_t n, int lane)
{
    int32_t bits = 0;
    for (int32_t i = lane; i < n; i += 64) {
        bits += t.numberOfBits[in[i]];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bits += __shfl_xor(bits, off);
    }
    return bits;
}

// HuffmanCompressor.compressSingleStream :88-134 executed by ONE lane (the others idle in this call).  The literals come 16 bytes per load,
// the next load requested before the current 16 symbols are encoded (one lane encoding from memory 4 bytes at a time waited a memory
// latency per 4 symbols: ~3 ms per 128 KiB frame); the flushes store 8 bytes where that stays inside this stream (`streamBytes`, its exact
// size, computed beforehand: the next stream begins right behind it and is written by another lane at the same time).
__device__ int32_t huf_encode_stream(const HufCTable& t, uint8_t* base, int32_t outputAddress, int32_t outputSize, const uint8_t* in, int32_t inputSize, int32_t streamBytes)
{
    if (outputSize < 8) {
        return 0;
    }
    BitOut bs;
    bo_init(bs, base, outputAddress, outputSize);
    const int32_t safeEnd = outputAddress + streamBytes;
    int32_t n = inputSize & ~3;
#define ZC_ENC(sym) bo_add_fast(bs, t.values[(sym)], t.numberOfBits[(sym)])
#define ZC_ENC4(w)            \
    ZC_ENC((w) >> 24);          \
    ZC_ENC(((w) >> 16) & 0xFF); \
    ZC_ENC(((w) >> 8) & 0xFF);  \
    ZC_ENC((w) & 0xFF);         \
    bo_flush_wide(bs, safeEnd);
    switch (inputSize & 3) {
        case 3: ZC_ENC(in[n + 2]);  // fallthrough
        case 2: ZC_ENC(in[n + 1]);  // fallthrough
        case 1:
            ZC_ENC(in[n + 0]);
            bo_flush_wide(bs, safeEnd);  // fallthrough
        default: break;
    }
    if (n >= 16) {
        u32x4 next = ld16(in + n - 16);
        while (n >= 16) {
            const u32x4 w = next;
            n -= 16;
            if (n >= 16) {
                next = ld16(in + n - 16);
            }
            ZC_ENC4(w.w);
            ZC_ENC4(w.z);
            ZC_ENC4(w.y);
            ZC_ENC4(w.x);
        }
    }
    for (; n > 0; n -= 4) {
        const uint32_t w = ld4(in + n - 4);
        ZC_ENC4(w);
    }
#undef ZC_ENC4
#undef ZC_ENC
    return bo_close(bs);
}

// size compressSingleStream would return for a stream of `bits` code bits in a buffer of outputSize bytes
__device__ __forceinline__ int32_t huf_stream_size(int32_t bits, int32_t outputSize)
{
    if (outputSize < 8) {
        return 0;
    }
    const int32_t total = bits + 1;  // end mark
    if ((total >> 3) >= outputSize - 8) {
        return 0;  // BitOutputStream.close(): currentAddress >= outputLimit
    }
    return (total + 7) >> 3;
}

// ---- literals section: encodeLiterals :262-378 ---------------------------------------------------------------
__device__ int32_t raw_literals(Ctx& c, uint8_t* base, int32_t outputAddress, int32_t outputSize, const uint8_t* in, int32_t inputSize)  // :407-431
{
    int32_t headerSize = 1;
    if (inputSize >= 32) headerSize++;
    if (inputSize >= 4096) headerSize++;
    ZC_CHECK(c, inputSize + headerSize <= outputSize);
    if (headerSize == 1) {
        base[outputAddress] = (uint8_t)(0 | (inputSize << 3));
    }
    else if (headerSize == 2) {
        st2(base + outputAddress, (uint32_t)(0 | (1 << 2) | (inputSize << 4)));
    }
    else {
        st_le(base + outputAddress, (uint32_t)(0 | (3 << 2) | (inputSize << 4)), 3);
    }
    ZC_CHECK(c, inputSize + 1 <= outputSize);
    wave_mem_order();
    group_copy<64>(base + outputAddress + headerSize, in, inputSize, c.lane);
    wave_mem_order();
    return headerSize + inputSize;
}

__device__ int32_t encode_literals(Ctx& c, Shared& sh, uint8_t* base, int32_t outputAddress, int32_t outputSize, const uint8_t* literals, int32_t literalsSize)
{
    if (literalsSize <= 63) {
        return raw_literals(c, base, outputAddress, outputSize, literals, literalsSize);
    }
    const int32_t headerSize = 3 + (literalsSize >= 1024 ? 1 : 0) + (literalsSize >= 16384 ? 1 : 0);
    ZC_CHECK(c, headerSize + 1 <= outputSize);
    wave_histogram(sh, literals, literalsSize, 256, c.lane);
    const int32_t maxSymbol = find_max_symbol(sh, 255);
    const int32_t largestCount = find_largest_count(sh, maxSymbol);
    if (largestCount == literalsSize) {  // rleLiterals :380-398
        const int32_t hs = 1 + (literalsSize > 31 ? 1 : 0) + (literalsSize > 4095 ? 1 : 0);
        if (hs == 1) {
            base[outputAddress] = (uint8_t)(1 | (literalsSize << 3));
        }
        else if (hs == 2) {
            st2(base + outputAddress, (uint32_t)(1 | (1 << 2) | (literalsSize << 4)));
        }
        else {
            st_le(base + outputAddress, (uint32_t)(1 | (3 << 2) | (literalsSize << 4)), 3);  // putInt; byte 3 is then overwritten
        }
        base[outputAddress + hs] = literals[0];
        return hs + 1;
    }
    else if (largestCount <= (int32_t)((uint32_t)literalsSize >> 7) + 4) {
        return raw_literals(c, base, outputAddress, outputSize, literals, literalsSize);
    }
    HufCTable& previousTable = sh.huf[c.previousTable];
    const bool canReuse = huf_table_is_valid(sh, previousTable, maxSymbol);
    const bool preferReuse = literalsSize <= 1024;
    HufCTable* table;
    int32_t serializedTableSize;
    bool reuseTable;
    if (preferReuse && canReuse) {
        table = &previousTable;
        reuseTable = true;
        serializedTableSize = 0;
    }
    else {
        c.previousCandidate = c.temporaryTable;  // borrowTemporaryTable
        c.temporaryCandidate = c.previousTable;
        HufCTable& newTable = sh.huf[c.temporaryTable];
        huf_table_initialize(sh, newTable, maxSymbol, huf_optimal_number_of_bits(11, literalsSize, maxSymbol));
        serializedTableSize = huf_table_write(c, sh, newTable, base, outputAddress + headerSize, outputSize - headerSize);
        ZC_PROPAGATE(serializedTableSize);
        if (canReuse && huf_estimate_compressed_size(sh, previousTable, maxSymbol) <= serializedTableSize + huf_estimate_compressed_size(sh, newTable, maxSymbol)) {
            table = &previousTable;
            reuseTable = true;
            serializedTableSize = 0;
            c.previousCandidate = c.previousTable;  // discardTemporaryTable
            c.temporaryCandidate = c.temporaryTable;
        }
        else {
            table = &newTable;
            reuseTable = false;
        }
    }
    // ---- HuffmanCompressor: sizes first (wave reduction), then one lane per stream ----
    const int32_t streamsAddress = outputAddress + headerSize + serializedTableSize;
    const int32_t streamsSize = outputSize - headerSize - serializedTableSize;
    const bool singleStream = literalsSize < 256;
    int32_t compressedSize;
    wave_mem_order();
    if (singleStream) {
        const int32_t bits = wave_code_bits(*table, literals, literalsSize, c.lane);
        compressedSize = huf_stream_size(bits, streamsSize);
        if (compressedSize != 0 && c.lane == 0) {
            huf_encode_stream(*table, base, streamsAddress, streamsSize, literals, literalsSize, compressedSize);
        }
    }
    else {  // compress4streams :26-86
        const int32_t segmentSize = (literalsSize + 3) / 4;
        compressedSize = 0;
        if (!(streamsSize < 6 + 1 + 1 + 1 + 8) && !(literalsSize <= 6 + 1 + 1 + 1)) {
            int32_t segStart[4], segLen[4], outStart[4], outAvail[4], segBytes[4];
            int32_t output = streamsAddress + 6;
            const int32_t outputLimit = streamsAddress + streamsSize;
            bool ok = true;
            for (int k = 0; k < 4; k++) {
                segStart[k] = k * segmentSize;
                segLen[k] = k < 3 ? segmentSize : literalsSize - 3 * segmentSize;
                outStart[k] = output;
                outAvail[k] = outputLimit - output;
                const int32_t bits = wave_code_bits(*table, literals + segStart[k], segLen[k], c.lane);
                segBytes[k] = ok ? huf_stream_size(bits, outAvail[k]) : 0;
                if (segBytes[k] == 0) {
                    ok = false;
                }
                output += segBytes[k];
            }
            if (ok) {
                if (c.lane < 4) {
                    int32_t mStart = 0, mLen = 0, mOut = 0, mAvail = 0, mBytes = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (c.lane == k) {
                            mStart = segStart[k];
                            mLen = segLen[k];
                            mOut = outStart[k];
                            mAvail = outAvail[k];
                            mBytes = segBytes[k];
                        }
                    }
                    huf_encode_stream(*table, base, mOut, mAvail, literals + mStart, mLen, mBytes);
                }
                st2(base + streamsAddress, (uint32_t)segBytes[0]);
                st2(base + streamsAddress + 2, (uint32_t)segBytes[1]);
                st2(base + streamsAddress + 4, (uint32_t)segBytes[2]);
                compressedSize = output - streamsAddress;
            }
        }
    }
    wave_mem_order();
    const int32_t totalSize = serializedTableSize + compressedSize;
    const int32_t minimumGain = (int32_t)((uint32_t)literalsSize >> 6) + 2;
    if (compressedSize == 0 || totalSize >= literalsSize - minimumGain) {
        c.previousCandidate = c.previousTable;
        c.temporaryCandidate = c.temporaryTable;
        return raw_literals(c, base, outputAddress, outputSize, literals, literalsSize);
    }
    const int32_t encodingType = reuseTable ? 3 : 2;
    if (headerSize == 3) {
        st_le(base + outputAddress, (uint32_t)(encodingType | ((singleStream ? 0 : 1) << 2) | (literalsSize << 4) | (totalSize << 14)), 3);
    }
    else if (headerSize == 4) {
        st4(base + outputAddress, (uint32_t)(encodingType | (2 << 2) | (literalsSize << 4) | (totalSize << 18)));
    }
    else {
        st4(base + outputAddress, (uint32_t)encodingType | (3u << 2) | ((uint32_t)literalsSize << 4) | ((uint32_t)totalSize << 22));
        base[outputAddress + 4] = (uint8_t)((uint32_t)totalSize >> 10);
    }
    return headerSize + totalSize;
}

// ---- sequences section: SequenceEncoder.compressSequences :66-209 ---------------------------------------------
__device__ __forceinline__ int32_t select_encoding_type(int32_t largestCount, int32_t sequenceCount, int32_t defaultLog, bool defaultAllowed)  // :299-341 (DFAST)
{
    if (largestCount == sequenceCount) {
        if (defaultAllowed && sequenceCount <= 2) {
            return 0;
        }
        return 1;
    }
    if (defaultAllowed) {
        const int64_t minNumberOfSequences = ((1LL << defaultLog) * 9) >> 3;
        if ((sequenceCount < minNumberOfSequences) || (largestCount < (sequenceCount >> (defaultLog - 1)))) {
            return 0;
        }
    }
    return 2;
}

// buildCompressionTable :211-226 ; returns bytes written or -1
__device__ int32_t build_compression_table(Ctx& c, Shared& sh, FseCTable& table, uint8_t* base, int32_t output, int64_t outputLimit, int32_t sequenceCount, int32_t maxTableLog,
                                           const uint8_t* codes, int32_t maxSymbol)
{
    const int32_t tableLog = fse_optimal_table_log(maxTableLog, sequenceCount, maxSymbol);
    const int32_t lastCode = codes[sequenceCount - 1];
    if (sh.counts[lastCode] > 1) {
        sh.counts[lastCode] = sh.counts[lastCode] - 1;
        sequenceCount--;
    }
    fse_normalize_counts(sh.norm, tableLog, sh.counts, sequenceCount, maxSymbol, c.lane);
    fse_initialize(sh, table, sh.norm, maxSymbol, tableLog);
    return fse_write_normalized_counts(c, sh, base, output, (int32_t)(outputLimit - output), sh.norm, maxSymbol, tableLog);
}

__device__ int32_t compress_sequences(Ctx& c, Shared& sh, uint8_t* base, int32_t outputAddress, int32_t outputSize)
{
    int32_t output = outputAddress;
    const int32_t outputLimit = outputAddress + outputSize;
    ZC_CHECK(c, outputLimit - output > 3 + 1);
    const int32_t sequenceCount = c.sequenceCount;
    if (sequenceCount < 0x7F) {
        base[output] = (uint8_t)sequenceCount;
        output++;
    }
    else if (sequenceCount < 0x7F00) {
        base[output] = (uint8_t)((uint32_t)sequenceCount >> 8 | 0x80);
        base[output + 1] = (uint8_t)sequenceCount;
        output += 2;
    }
    else {
        base[output] = 0xFF;
        output++;
        st2(base + output, (uint32_t)(sequenceCount - 0x7F00));
        output += 2;
    }
    if (sequenceCount == 0) {
        return output - outputAddress;
    }
    const int32_t headerAddress = output++;
    int32_t maxSymbol, largestCount;

    // literal lengths
    wave_histogram(sh, c.codeLL, sequenceCount, 53, c.lane);
    maxSymbol = find_max_symbol(sh, 35);
    largestCount = find_largest_count(sh, maxSymbol);
    const int32_t llType = select_encoding_type(largestCount, sequenceCount, 6, true);
    const FseCTable* llTable;
    if (llType == 1) {
        base[output] = c.codeLL[0];
        output++;
        fse_init_rle(sh.ll, maxSymbol);
        llTable = &sh.ll;
    }
    else if (llType == 0) {
        llTable = &sh.dflt[0];
    }
    else {
        const int32_t n = build_compression_table(c, sh, sh.ll, base, output, outputLimit, sequenceCount, 9, c.codeLL, maxSymbol);
        ZC_PROPAGATE(n);
        output += n;
        llTable = &sh.ll;
    }

    // offsets
    wave_histogram(sh, c.codeOF, sequenceCount, 53, c.lane);
    maxSymbol = find_max_symbol(sh, 31);
    largestCount = find_largest_count(sh, maxSymbol);
    const int32_t ofType = select_encoding_type(largestCount, sequenceCount, 5, maxSymbol < 28);
    const FseCTable* ofTable;
    if (ofType == 1) {
        base[output] = c.codeOF[0];
        output++;
        fse_init_rle(sh.of, maxSymbol);
        ofTable = &sh.of;
    }
    else if (ofType == 0) {
        ofTable = &sh.dflt[1];
    }
    else {  // the Java code passes output + outputSize as the limit here (:155)
        const int32_t n = build_compression_table(c, sh, sh.of, base, output, (int64_t)output + outputSize, sequenceCount, 8, c.codeOF, maxSymbol);
        ZC_PROPAGATE(n);
        output += n;
        ofTable = &sh.of;
    }

    // match lengths
    wave_histogram(sh, c.codeML, sequenceCount, 53, c.lane);
    maxSymbol = find_max_symbol(sh, 52);
    largestCount = find_largest_count(sh, maxSymbol);
    const int32_t mlType = select_encoding_type(largestCount, sequenceCount, 6, true);
    const FseCTable* mlTable;
    if (mlType == 1) {
        base[output] = c.codeML[0];
        output++;
        fse_init_rle(sh.ml, maxSymbol);
        mlTable = &sh.ml;
    }
    else if (mlType == 0) {
        mlTable = &sh.dflt[2];
    }
    else {
        const int32_t n = build_compression_table(c, sh, sh.ml, base, output, outputLimit, sequenceCount, 9, c.codeML, maxSymbol);
        ZC_PROPAGATE(n);
        output += n;
        mlTable = &sh.ml;
    }
    base[headerAddress] = (uint8_t)((llType << 6) | (ofType << 4) | (mlType << 2));

    // encodeSequences :228-297.  The stream is a concatenation of bit fields, least significant first: per sequence, from the last to the first,
    // the state bits of offset, match length and literal length (fse_encode :126-131), then the extra bits of literal length, match length and
    // offset; behind them the three final states and the end mark.  Only the three STATE CHAINS are serial, and they are independent of each
    // other: lanes 0 / 1 / 2 walk the offset / match-length / literal-length chain side by side (one instruction stream, a table lookup per
    // step and chain) and leave every step's {bits, count} in LDS; then all 64 lanes put their sequences' fields together, a scan of the bit
    // counts places them, and the words go out whole.  Until round 4 one loop did all of it, sequence after sequence, with every lane
    // computing the same values (~0.27 us per sequence: half the entropy kernel on text).  What the Java flushes do along the way -- which bytes
    // are stored when -- is not observable: the result is the same bytes, and the one way the stream can fail (it does not fit: close()
    // returns 0, :82-92) is decided by the same comparison of its final position with the buffer's limit.
    ZC_CHECK(c, outputLimit - output >= 8);
    const int lane = c.lane;  // (a local: `c` lives in memory, and a store to LDS may be a store to it for all the compiler knows)
    const int32_t sStart = output;
    const int32_t n = sequenceCount - 1;
    const int chain = lane < 2 ? lane : 2;
    const FseCTable* const myTable = chain == 0 ? ofTable : (chain == 1 ? mlTable : llTable);
    int32_t st = fse_begin(*myTable, chain == 0 ? c.codeOF[n] : (chain == 1 ? c.codeML[n] : c.codeLL[n]));
    uint32_t* const win = (uint32_t*)sh.cumulative;  // the stream's words under construction: window[0] is word P >> 5 of the stream (258 words; a group of 64 sequences takes at most 182)
    int32_t* const piece = sh.nodeCount;             // [64][4]: the chains' {bits | count << 16} of a group's steps
    static_assert(sizeof(sh.cumulative) >= 4 * 192 && sizeof(sh.nodeCount) >= 4 * 256, "the packing window and the chains' pieces fit the tree builder's arrays");
    for (int w = lane; w < 192; w += 64) {
        win[w] = 0;
    }
    wave_sync();
    // `value`'s low `bits` bits (bits <= 31) appended to a lane's 128-bit field (lo, hi, nb)
    auto append = [](uint64_t& lo, uint64_t& hi, int32_t& nb, int32_t value, int32_t bits) {
        const uint64_t x = (uint64_t)((uint32_t)value & (uint32_t)((1ull << bits) - 1ull));
        lo |= nb < 64 ? x << (nb & 63) : 0ull;
        hi |= nb >= 64 ? x << ((nb - 64) & 63) : (nb + bits > 64 ? x >> ((64 - nb) & 63) : 0ull);
        nb += bits;
    };
    // a field of at most 96 bits ORed into the window at bit `at` (< 32 + 64 x 90)
    auto put = [&](uint64_t lo, uint64_t hi, int32_t nb, int32_t at) {
        if (nb <= 0) {
            return;
        }
        const int32_t w0 = at >> 5, sh5 = at & 31;
        const uint32_t v0 = (uint32_t)lo, v1 = (uint32_t)(lo >> 32), v2 = (uint32_t)hi, v3 = (uint32_t)(hi >> 32);
        const uint32_t o0 = v0 << sh5;
        const uint32_t o1 = sh5 == 0 ? v1 : ((v1 << sh5) | (v0 >> (32 - sh5)));
        const uint32_t o2 = sh5 == 0 ? v2 : ((v2 << sh5) | (v1 >> (32 - sh5)));
        const uint32_t o3 = sh5 == 0 ? v3 : ((v3 << sh5) | (v2 >> (32 - sh5)));
        const uint32_t o4 = sh5 == 0 ? 0u : (v3 >> (32 - sh5));
        if (o0 != 0) atomicOr(win + w0, o0);
        if (o1 != 0) atomicOr(win + w0 + 1, o1);
        if (o2 != 0) atomicOr(win + w0 + 2, o2);
        if (o3 != 0) atomicOr(win + w0 + 3, o3);
        if (o4 != 0) atomicOr(win + w0 + 4, o4);
    };
    int32_t P = 0;      // (uniform) bits of the stream so far
    int32_t firstBits = 0;
    bool ovf = false;   // (uniform) the stream has left its buffer: nothing more is stored, close() will say so
    {
        // the last sequence's extra bits open the stream (:241-245)
        uint64_t lo = 0, hi = 0;
        int32_t nb = 0;
        append(lo, hi, nb, c.seqLitLen[n], LL_BITS[c.codeLL[n]]);
        append(lo, hi, nb, c.seqMatchLen[n], ML_BITS[c.codeML[n]]);
        append(lo, hi, nb, c.seqOffset[n], c.codeOF[n]);
        if (lane == 0) {
            put(lo, hi, nb, 0);
        }
        firstBits = nb;
    }
    wave_sync();
    // whole words of the window leave for the stream; the word under construction moves to the window's start
    auto flush_words = [&](int32_t newP) {
        const int32_t w0 = P >> 5, whole = (newP >> 5) - w0;
        if ((int64_t)sStart + 4LL * (w0 + whole) > (int64_t)outputLimit) {
            ovf = true;
        }
        if (!ovf) {
            for (int32_t w = lane; w < whole; w += 64) {
                st4(base + sStart + 4 * (w0 + w), win[w]);
            }
        }
        const uint32_t carry = win[whole];
        wave_sync();
        for (int32_t w = lane; w <= whole; w += 64) {
            win[w] = w == 0 ? carry : 0u;
        }
        wave_sync();
        P = newP;
    };
    flush_words(firstBits);
    // The sequences 64 at a time: lane l fetches sequence top - l -- its codes, its three values, the bit counts of its codes and, from the
    // three tables, the two per-symbol deltas of each code (none of which depends on the states).
    for (int32_t top = sequenceCount - 2; top >= 0; top -= 64) {
        const int32_t mine = top - lane;
        int32_t vLL = 0, vML = 0, vOF = 0, bLL = 0, bML = 0, bOF = 0, nLL = 0, fLL = 0, nML = 0, fML = 0, nOF = 0, fOF = 0;
        if (mine >= 0) {
            const int32_t llCode = c.codeLL[mine];
            const int32_t ofCode = c.codeOF[mine];
            const int32_t mlCode = c.codeML[mine];
            vLL = c.seqLitLen[mine];
            vML = c.seqMatchLen[mine];
            vOF = c.seqOffset[mine];
            bLL = LL_BITS[llCode];
            bML = ML_BITS[mlCode];
            bOF = ofCode;
            nLL = llTable->deltaNumberOfBits[llCode];
            fLL = llTable->deltaFindState[llCode];
            nML = mlTable->deltaNumberOfBits[mlCode];
            fML = mlTable->deltaFindState[mlCode];
            nOF = ofTable->deltaNumberOfBits[ofCode];
            fOF = ofTable->deltaFindState[ofCode];
        }
        const int count = top + 1 < 64 ? top + 1 : 64;
        // ---- the chains: step k is sequence top - k ----
        for (int k = 0; k < count; k++) {  // (uniform)
#define ZC_RL(v) __builtin_amdgcn_readlane((v), k)
            const int32_t dnOF = ZC_RL(nOF), dnML = ZC_RL(nML), dnLL = ZC_RL(nLL), dfOF = ZC_RL(fOF), dfML = ZC_RL(fML), dfLL = ZC_RL(fLL);  // (lane reads: every lane takes part)
            const int32_t dn = chain == 0 ? dnOF : (chain == 1 ? dnML : dnLL);
            const int32_t df = chain == 0 ? dfOF : (chain == 1 ? dfML : dfLL);
#undef ZC_RL
            const int32_t ob = (int32_t)((uint32_t)(st + dn) >> 16);  // fse_encode :126-131
            if (lane < 3) {
                piece[4 * k + lane] = (int32_t)(((uint32_t)st & ((1u << (ob & 31)) - 1u)) | ((uint32_t)ob << 16));
            }
            st = myTable->nextState[(int32_t)((uint32_t)st >> (ob & 31)) + df];
        }
        wave_sync();
        // ---- the fields of sequence top - lane: state bits of offset, match length, literal length; extra bits of literal length, match length, offset ----
        uint64_t lo = 0, hi = 0;
        int32_t nb = 0;
        if (lane < count) {
            const int32_t p0 = piece[4 * lane], p1 = piece[4 * lane + 1], p2 = piece[4 * lane + 2];
            append(lo, hi, nb, p0 & 0xFFFF, p0 >> 16);
            append(lo, hi, nb, p1 & 0xFFFF, p1 >> 16);
            append(lo, hi, nb, p2 & 0xFFFF, p2 >> 16);
            append(lo, hi, nb, vLL, bLL);
            append(lo, hi, nb, vML, bML);
            append(lo, hi, nb, vOF, bOF);
        }
        const int32_t incl = sx::wave_scan_incl(nb, lane);
        const int32_t groupBits = sx::wave_bcast(incl, 63);
        put(lo, hi, nb, (P & 31) + incl - nb);
        wave_sync();
        flush_words(P + groupBits);
    }
    // the stream goes on through the Java-shaped writer: its position, the bits of its last unfinished byte
    BitOut bs;
    bo_init(bs, base, sStart, outputLimit - sStart);
    {
        const uint32_t carry = win[0];
        const int32_t wordStart = 4 * (P >> 5), byteAt = P >> 3;
        if (!ovf && byteAt > wordStart && lane == 0) {
            st_le(base + sStart + wordStart, (uint64_t)carry, byteAt - wordStart);
        }
        bs.current = sStart + byteAt;
        bs.bitCount = P & 7;
        bs.container = (uint64_t)(carry >> (8 * (byteAt - wordStart))) & ((1ull << (P & 7)) - 1ull);
        if (ovf || bs.current > bs.limit) {
            bs.current = bs.limit;
        }
    }
    wave_sync();
    const int32_t ofState = __builtin_amdgcn_readlane(st, 0), mlState = __builtin_amdgcn_readlane(st, 1), llState = __builtin_amdgcn_readlane(st, 2);
    fse_finish(*mlTable, bs, mlState);
    fse_finish(*ofTable, bs, ofState);
    fse_finish(*llTable, bs, llState);
    const int32_t streamSize = bo_close(bs);
    ZC_CHECK(c, streamSize > 0);
    output += streamSize;
    return output - outputAddress;
}

// ---- match finder: DoubleFastBlockCompressor.compressBlock :28-180 ---------------------------------------------
__device__ __forceinline__ int32_t hash4(uint32_t v, int32_t bits) { return (int32_t)((v * 0x9E3779B1u) >> (32 - bits)); }
__device__ __forceinline__ int32_t hash5(uint64_t v, int32_t bits) { return (int32_t)(((v << 24) * 0xCF1BBCDCBBULL) >> (64 - bits)); }
__device__ __forceinline__ int32_t hash8(uint64_t v, int32_t bits) { return (int32_t)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - bits)); }
__device__ __forceinline__ int32_t hash_short(const Ctx& c, int32_t address)  // hash(...) :213-222 ; level 3 only uses search lengths 4 and 5
{
    return c.searchLength == 5 ? hash5(ld8(c.in + address), c.chainLog) : hash4(ld4(c.in + address), c.chainLog);
}

__device__ __forceinline__ void store_sequence(Ctx& c, int32_t literalAddress, int32_t literalLength, int32_t offsetCode, int32_t matchLengthBase)  // SequenceStore.storeSequence :83-111
{
    wave_mem_order();
    group_copy<64>(c.litBuf + c.literalsLength, c.in + literalAddress, literalLength, c.lane);
    c.literalsLength += literalLength;
    const int32_t i = c.sequenceCount;
    if (literalLength > 65535) {
        c.longLengthField = 1;
        c.longLengthPosition = i;
    }
    c.seqLitLen[i] = literalLength;
    c.seqOffset[i] = offsetCode + 1;
    if (matchLengthBase > 65535) {
        c.longLengthField = 2;
        c.longLengthPosition = i;
    }
    c.seqMatchLen[i] = matchLengthBase;
    c.sequenceCount = i + 1;
}

// for every lane, the mask of lanes whose `key` (low `bits` bits) equals its own
__device__ __forceinline__ unsigned long long zc_match_any(uint32_t key, int bits, unsigned long long active)
{
    unsigned long long eq = active;
    for (int b = 0; b < bits; b++) {
        const bool bit = (key >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        eq &= bit ? m : ~m;
    }
    return eq;
}

__device__ int32_t dfast_compress_block(Ctx& c, int32_t inputAddress, int32_t inputSize)
{
    const uint8_t* __restrict__ in = c.in;
    const int32_t windowBase = c.windowBaseOffset;
    int32_t* longTable = c.hashTable;
    int32_t* shortTable = c.chainTable;
    const int32_t longBits = c.hashLog;
    const int32_t inputEnd = inputAddress + inputSize;
    const int32_t inputLimit = inputEnd - 8;
    int32_t input = inputAddress;
    int32_t anchor = inputAddress;
    int32_t offset1 = c.offset0;
    int32_t offset2 = c.offset1;
    int32_t savedOffset = 0;
    if (input - windowBase == 0) {
        input++;
    }
    const int32_t maxRep = input - windowBase;
    if (offset2 > maxRep) {
        savedOffset = offset2;
        offset2 = 0;
    }
    if (offset1 > maxRep) {
        savedOffset = offset1;
        offset1 = 0;
    }
    int32_t width = 8;  // lanes of the next batch probe: narrow right after a match (text finds the next one within a few positions), 64 once a batch found nothing
    while (input < inputLimit) {
        uint64_t here = 0;
        int32_t shortHash = 0, longHash = 0, shortMatch = 0, longMatch = 0;
        bool repHit = false, longHit = false, shortHit = false;
        bool probed = false;
        if (c.batchProbe && input - anchor < 256 - 64) {
            // Batch probe: while nothing matches the serial loop visits input, input + 1, ... (its step is
            // ((input - anchor) >> 8) + 1 = 1 here) and changes nothing but the two tables, so the lanes test the next
            // `width` positions at once -- three dependent HBM round trips per batch instead of per position.  A lane
            // sees the tables as the serial loop would: the latest earlier lane with the same hash, else the stored
            // entry.  Lanes before the first candidate match commit their inserts (latest position per slot); the
            // first candidate's position continues below with what its lane has already loaded.
            const int lane = c.lane;
            const int32_t p = input + lane;
            const bool act = lane < width && p < inputLimit;
            const unsigned long long actMask = __ballot(act);
            const uint64_t hereB = act ? ld8(in + p) : 0ull;
            const int32_t sh = c.searchLength == 5 ? hash5(hereB, c.chainLog) : hash4((uint32_t)hereB, c.chainLog);
            const int32_t lh = hash8(hereB, longBits);
            int32_t sm = act ? shortTable[sh] : 0;
            int32_t lm = act ? longTable[lh] : 0;
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned long long sameS = zc_match_any((uint32_t)sh, c.chainLog, actMask);
            const unsigned long long sameL = zc_match_any((uint32_t)lh, longBits, actMask);
            if (act && (sameS & below) != 0) {
                sm = input + (63 - __builtin_clzll(sameS & below));
            }
            if (act && (sameL & below) != 0) {
                lm = input + (63 - __builtin_clzll(sameL & below));
            }
            bool r = false, l = false, sHit = false;
            if (act) {
                r = offset1 > 0 && ld4(in + p + 1 - offset1) == (uint32_t)(hereB >> 8);
                l = lm > windowBase && ld8(in + lm) == hereB;
                sHit = sm > windowBase && ld4(in + sm) == (uint32_t)hereB;
            }
            const unsigned long long hitMask = __ballot(r || l || sHit);
            const unsigned long long stop = hitMask | ~actMask;
            const int first = stop != 0 ? __builtin_ctzll(stop) : 64;
            const unsigned long long upTo = first >= 64 ? ~0ull : ((1ull << first) - 1ull);  // lanes that found nothing
            const unsigned long long above = lane >= 63 ? 0ull : ~((2ull << lane) - 1ull);
            if (lane < first) {
                if ((sameL & upTo & above) == 0) {
                    longTable[lh] = p;
                }
                if ((sameS & upTo & above) == 0) {
                    shortTable[sh] = p;
                }
            }
            wave_mem_order();
            input += first;
            if (first >= 64 || ((hitMask >> first) & 1ull) == 0) {
                width = 64;  // nothing in this batch (or the end of the block): keep going, wide
                continue;
            }
            here = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(hereB >> 32), first) << 32) | (uint32_t)__shfl((int)(uint32_t)hereB, first);
            shortHash = __shfl(sh, first);
            longHash = __shfl(lh, first);
            shortMatch = __shfl(sm, first);
            longMatch = __shfl(lm, first);
            repHit = __shfl(r ? 1 : 0, first) != 0;
            longHit = __shfl(l ? 1 : 0, first) != 0;
            shortHit = __shfl(sHit ? 1 : 0, first) != 0;
            probed = true;
        }
        if (!probed) {
            here = ld8(in + input);
            shortHash = c.searchLength == 5 ? hash5(here, c.chainLog) : hash4((uint32_t)here, c.chainLog);
            shortMatch = shortTable[shortHash];
            longHash = hash8(here, longBits);
            longMatch = longTable[longHash];
            repHit = offset1 > 0 && ld4(in + input + 1 - offset1) == ld4(in + input + 1);
            if (!repHit) {
                longHit = longMatch > windowBase && ld8(in + longMatch) == here;
                if (!longHit) {
                    shortHit = shortMatch > windowBase && ld4(in + shortMatch) == (uint32_t)here;
                }
            }
        }
        const int32_t current = input;
        longTable[longHash] = current;
        shortTable[shortHash] = current;
        int32_t matchLength;
        int32_t offset;
        if (repHit) {
            matchLength = wave_count(in, input + 1 + 4, input + 1 + 4 - offset1, inputEnd, c.lane) + 4;
            input++;
            store_sequence(c, anchor, input - anchor, 0, matchLength - 3);
        }
        else {
            if (longHit) {
                matchLength = wave_count(in, input + 8, longMatch + 8, inputEnd, c.lane) + 8;
                offset = input - longMatch;
                while (input > anchor && longMatch > windowBase && in[input - 1] == in[longMatch - 1]) {
                    input--;
                    longMatch--;
                    matchLength++;
                }
            }
            else if (shortHit) {
                const uint64_t next = ld8(in + input + 1);
                const int32_t nextHash = hash8(next, longBits);
                int32_t nextMatch = longTable[nextHash];
                longTable[nextHash] = current + 1;
                if (nextMatch > windowBase && ld8(in + nextMatch) == next) {
                    matchLength = wave_count(in, input + 1 + 8, nextMatch + 8, inputEnd, c.lane) + 8;
                    input++;
                    offset = input - nextMatch;
                    while (input > anchor && nextMatch > windowBase && in[input - 1] == in[nextMatch - 1]) {
                        input--;
                        nextMatch--;
                        matchLength++;
                    }
                }
                else {
                    matchLength = wave_count(in, input + 4, shortMatch + 4, inputEnd, c.lane) + 4;
                    offset = input - shortMatch;
                    while (input > anchor && shortMatch > windowBase && in[input - 1] == in[shortMatch - 1]) {
                        input--;
                        shortMatch--;
                        matchLength++;
                    }
                }
            }
            else {
                input += ((input - anchor) >> 8) + 1;
                continue;
            }
            offset2 = offset1;
            offset1 = offset;
            store_sequence(c, anchor, input - anchor, offset + 2, matchLength - 3);
        }
        width = input - anchor >= 24 ? 64 : 8;  // the literal run just closed predicts the next one: wide batches for long runs, narrow for text
        input += matchLength;
        anchor = input;
        if (input <= inputLimit) {
            const uint64_t a = ld8(in + current + 2);
            longTable[hash8(a, longBits)] = current + 2;
            shortTable[c.searchLength == 5 ? hash5(a, c.chainLog) : hash4((uint32_t)a, c.chainLog)] = current + 2;
            const uint64_t b = ld8(in + input - 2);
            longTable[hash8(b, longBits)] = input - 2;
            shortTable[c.searchLength == 5 ? hash5(b, c.chainLog) : hash4((uint32_t)b, c.chainLog)] = input - 2;
            while (input <= inputLimit && offset2 > 0 && ld4(in + input) == ld4(in + input - offset2)) {
                const int32_t repetitionLength = wave_count(in, input + 4, input + 4 - offset2, inputEnd, c.lane) + 4;
                const int32_t temp = offset2;
                offset2 = offset1;
                offset1 = temp;
                const uint64_t r = ld8(in + input);
                shortTable[c.searchLength == 5 ? hash5(r, c.chainLog) : hash4((uint32_t)r, c.chainLog)] = input;
                longTable[hash8(r, longBits)] = input;
                store_sequence(c, anchor, 0, 0, repetitionLength - 3);
                input += repetitionLength;
                anchor = input;
            }
        }
    }
    c.tempOffset0 = offset1 != 0 ? offset1 : savedOffset;
    c.tempOffset1 = offset2 != 0 ? offset2 : savedOffset;
    return inputEnd - anchor;
}

#include "zstd_dfast_mw.h"  // dfast_compress_block_mw (c.batchProbe == 2)

__device__ __forceinline__ int32_t match_finder(Ctx& c, int32_t inputAddress, int32_t inputSize)
{
    return c.batchProbe == 2 ? dfast_compress_block_mw(c, inputAddress, inputSize) : dfast_compress_block(c, inputAddress, inputSize);
}

// ---- blocks and frame: ZstdFrameCompressor :136-260 --------------------------------------------------------------
__device__ int32_t compress_block(Ctx& c, Shared& sh, int32_t inputAddress, int32_t inputSize, int32_t outputAddress, int32_t outputSize)
{
    if (inputSize < 3 + 3 + 1) {
        return 0;
    }
    const int32_t newOffset = (inputAddress + inputSize) - c.windowSize;  // enforceMaxDistance
    if (c.windowBaseOffset < newOffset) {
        c.windowBaseOffset = newOffset;
    }
    c.literalsLength = 0;
    c.sequenceCount = 0;
    c.longLengthField = 0;
    if (c.pre != nullptr) {
        c.sequenceCount = c.pre[1];
        c.literalsLength = c.pre[2];
        c.longLengthField = c.pre[3];
        c.longLengthPosition = c.pre[4];
        c.tempOffset0 = c.pre[5];
        c.tempOffset1 = c.pre[6];
    }
    else {
        const int32_t lastLiteralsSize = match_finder(c, inputAddress, inputSize);
        wave_mem_order();
        group_copy<64>(c.litBuf + c.literalsLength, c.in + inputAddress + inputSize - lastLiteralsSize, lastLiteralsSize, c.lane);
        c.literalsLength += lastLiteralsSize;
    }
    wave_mem_order();
    // generateCodes :121-135 : one sequence per lane per step
    for (int32_t i = c.lane; i < c.sequenceCount; i += 64) {
        const int32_t ll = c.seqLitLen[i];
        c.codeLL[i] = (uint8_t)(ll >= 64 ? highest_bit((uint32_t)ll) + 19 : LL_CODE[ll]);
        c.codeOF[i] = (uint8_t)highest_bit((uint32_t)c.seqOffset[i]);
        const int32_t ml = c.seqMatchLen[i];
        c.codeML[i] = (uint8_t)(ml >= 128 ? highest_bit((uint32_t)ml) + 36 : ML_CODE[ml]);
    }
    wave_mem_order();
    if (c.longLengthField == 1) {
        c.codeLL[c.longLengthPosition] = 35;
    }
    if (c.longLengthField == 2) {
        c.codeML[c.longLengthPosition] = 52;
    }
    wave_mem_order();
    const int32_t outputLimit = outputAddress + outputSize;
    int32_t output = outputAddress;
    const int32_t compressedLiteralsSize = encode_literals(c, sh, c.out, output, outputLimit - output, c.litBuf, c.literalsLength);
    ZC_PROPAGATE(compressedLiteralsSize);
    output += compressedLiteralsSize;
    const int32_t compressedSequencesSize = compress_sequences(c, sh, c.out, output, outputLimit - output);
    ZC_PROPAGATE(compressedSequencesSize);
    const int32_t compressedSize = compressedLiteralsSize + compressedSequencesSize;
    if (compressedSize == 0) {
        return 0;
    }
    const int32_t maxCompressedSize = inputSize - ((int32_t)((uint32_t)inputSize >> 6) + 2);
    if (compressedSize > maxCompressedSize) {
        return 0;
    }
    c.offset0 = c.tempOffset0;  // context.commit()
    c.offset1 = c.tempOffset1;
    c.temporaryTable = c.temporaryCandidate;
    c.previousTable = c.previousCandidate;
    return compressedSize;
}

// CompressionParameters.compute :256-299 (level 3)
__device__ __forceinline__ void compute_parameters(Ctx& c)
{
    const int32_t inputSize = c.inLen;
    const int table = inputSize <= 16 * 1024 ? 3 : (inputSize <= 128 * 1024 ? 2 : (inputSize <= 256 * 1024 ? 1 : 0));
    int32_t windowLog = LEVEL3[table][0], chainLog = LEVEL3[table][1], hashLog = LEVEL3[table][2];
    c.searchLength = LEVEL3[table][3];
    const int32_t inputSizeLog = inputSize < 64 ? 6 : highest_bit((uint32_t)(inputSize - 1)) + 1;
    if (windowLog > inputSizeLog) {
        windowLog = inputSizeLog;
    }
    if (hashLog > windowLog + 1) {
        hashLog = windowLog + 1;
    }
    if (chainLog > windowLog) {
        chainLog -= (chainLog - windowLog);
    }
    if (windowLog < 10) {
        windowLog = 10;
    }
    c.windowLog = windowLog;
    c.windowSize = 1 << windowLog;
    c.blockSize = c.windowSize < MAX_BLOCK_SIZE ? c.windowSize : MAX_BLOCK_SIZE;
    c.chainLog = chainLog;
    c.hashLog = hashLog;
}

// Two-kernel path (inputs of one block, 7..128 KiB -- BASELINE configs[3] / [4]): the match finder is 84-98 % of the
// encoder's time and needs neither the entropy stage's LDS nor most of its registers, so it runs as its own kernel with
// more wavefronts per CU; its results (sequence store, literals, the two candidate repeat offsets) wait in the item's
// scratch for the entropy kernel.  Larger inputs stay in the one kernel: whether block k's repeat offsets are committed
// depends on block k's entropy outcome (ZstdFrameCompressor.java:246-258), so their match finding cannot run ahead.
// the match kernel's record (first 256 bytes of an item's scratch): valid, sequenceCount, literalsLength, longLengthField, longLengthPosition, tempOffset0, tempOffset1
__device__ __forceinline__ bool split_eligible(int32_t inLen) { return inLen >= 7 && inLen <= MAX_BLOCK_SIZE; }

__device__ int32_t zstd_compress_item(Ctx& c, Shared& sh)
{
    const int32_t inputSize = c.inLen;
    const int32_t outputLimit = c.outCap;
    int32_t output = 0;
    compute_parameters(c);
    ZC_CHECK(c, outputLimit - output >= 4);  // writeMagic :55-61
    st4(c.out + output, 0xFD2FB528u);
    output += 4;
    ZC_CHECK(c, outputLimit - output >= 14);  // writeFrameHeader :64-121
    {
        const int32_t contentSizeDescriptor = (inputSize >= 256 ? 1 : 0) + (inputSize >= 65536 + 256 ? 1 : 0);
        int32_t fhd = (contentSizeDescriptor << 6) | 0x04;
        const bool singleSegment = c.windowSize >= inputSize;
        if (singleSegment) {
            fhd |= 0x20;
        }
        c.out[output++] = (uint8_t)fhd;
        if (!singleSegment) {
            c.out[output++] = (uint8_t)((c.windowLog - 10) << 3);  // window sizes here are powers of two: mantissa 0
        }
        if (contentSizeDescriptor == 0) {
            if (singleSegment) {
                c.out[output++] = (uint8_t)inputSize;
            }
        }
        else if (contentSizeDescriptor == 1) {
            st2(c.out + output, (uint32_t)(inputSize - 256));
            output += 2;
        }
        else {
            st4(c.out + output, (uint32_t)inputSize);
            output += 4;
        }
    }
    // compressFrame :152-179 with a fresh CompressionContext
    {
        c.offset0 = 1;
        c.offset1 = 4;
        c.tempOffset0 = c.tempOffset1 = 0;
        c.windowBaseOffset = 0;
        if (c.pre == nullptr) {  // (with a precomputed record the match finder -- and its tables -- already ran elsewhere)
            wave_fill((uint8_t*)c.hashTable, 0, 4 << c.hashLog, c.lane);
            wave_fill((uint8_t*)c.chainTable, 0, 4 << c.chainLog, c.lane);
        }
        for (int i = c.lane; i < 256; i += 64) {
            sh.huf[0].numberOfBits[i] = 0;
            sh.huf[1].numberOfBits[i] = 0;
        }
        sh.huf[0].maxSymbol = 0;
        sh.huf[0].maxNumberOfBits = 0;
        sh.huf[1].maxSymbol = 0;
        sh.huf[1].maxNumberOfBits = 0;
        c.previousTable = 0;
        c.temporaryTable = 1;
        c.previousCandidate = 0;
        c.temporaryCandidate = 1;
        __syncthreads();
        wave_mem_order();
        int32_t blockSize = c.blockSize;
        int32_t outputSize = outputLimit - output;
        int32_t remaining = inputSize;
        int32_t input = 0;
        do {
            ZC_CHECK(c, outputSize >= 3 + 3);
            const bool lastBlock = blockSize >= remaining;
            blockSize = blockSize < remaining ? blockSize : remaining;
            int32_t compressedSize = 0;  // writeCompressedBlock :181-204
            if (blockSize > 0) {
                compressedSize = compress_block(c, sh, input, blockSize, output + 3, outputSize - 3);
                ZC_PROPAGATE(compressedSize);
            }
            if (compressedSize == 0) {
                ZC_CHECK(c, blockSize + 3 <= outputSize);
                st_le(c.out + output, (uint32_t)((lastBlock ? 1 : 0) | (0 << 1) | (blockSize << 3)), 3);
                wave_mem_order();
                group_copy<64>(c.out + output + 3, c.in + input, blockSize, c.lane);
                wave_mem_order();
                compressedSize = 3 + blockSize;
            }
            else {
                st_le(c.out + output, (uint32_t)((lastBlock ? 1 : 0) | (2 << 1) | (compressedSize << 3)), 3);
                compressedSize += 3;
            }
            input += blockSize;
            remaining -= blockSize;
            output += compressedSize;
            outputSize -= compressedSize;
        } while (remaining > 0);
    }
    ZC_CHECK(c, outputLimit - output >= 4);  // writeChecksum :123-134
    const uint64_t hash = wave_xxh64(c.in, inputSize, c.lane);
    st4(c.out + output, (uint32_t)hash);
    output += 4;
    return output;
}

}  // namespace zc

}  // namespace achip
