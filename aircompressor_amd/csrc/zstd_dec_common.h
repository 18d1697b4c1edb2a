// zstd_dec_common.h -- pieces of the Zstd decoder shared by zstd_decompress.hip (one wavefront per item, any
// input) and zstd_decompress_pipe.hip (five-stage pipeline for single-block frames): format constants, the
// backward bit reader, FSE table reading / building, Huffman table reading.  Reference lines are cited per function.
#pragma once
#include "achip_rings.h"

namespace achip {

namespace zd {
using ZRings = Rings<64, 2048, 4096>;  // one wavefront per item: 1 KiB refill / flush chunks, back-references within 3056 bytes served from LDS
constexpr int MAX_BLOCK_SIZE = 128 * 1024;
constexpr int MAX_WINDOW_SIZE = 1 << 23;
constexpr int HUF_MAX_TABLE_LOG = 12;
constexpr int SEQ_RING = 1024;
constexpr int LIT_SLAB = MAX_BLOCK_SIZE + 64;

static __constant__ int32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
static __constant__ int32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                                    35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 0x83, 0x103, 0x203, 0x403, 0x803, 0x1003, 0x2003, 0x4003, 0x8003, 0x10003};
static __constant__ int32_t OF_BASE[29] = {0, 1, 1, 5, 0xD, 0x1D, 0x3D, 0x7D, 0xFD, 0x1FD, 0x3FD, 0x7FD, 0xFFD, 0x1FFD, 0x3FFD, 0x7FFD, 0xFFFD, 0x1FFFD, 0x3FFFD, 0x7FFFD,
                                    0xFFFFD, 0x1FFFFD, 0x3FFFFD, 0x7FFFFD, 0xFFFFFD, 0x1FFFFFD, 0x3FFFFFD, 0x7FFFFFD, 0xFFFFFFD};
static __constant__ uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static __constant__ uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
// predefined distributions, RFC 8878 3.1.1.3.2.2 (they rebuild ZstdFrameDecompressor.java:85-113 exactly; checked entry by entry against the Java source in the CPU test suite)
static __constant__ int16_t LL_DEFAULT_NORM[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static __constant__ int16_t OF_DEFAULT_NORM[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static __constant__ int16_t ML_DEFAULT_NORM[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};

// FSE decoding table entry: newState (low 16, signed) | symbol << 16 | numberOfBits << 24
struct FseTable {
    uint32_t e[512];
};

// tables of the block being parsed: shared by the one-kernel decoder and by the pipeline's parse stage
struct TableShared {
    uint16_t huf[1 << HUF_MAX_TABLE_LOG];  // symbol | numberOfBits << 8
    FseTable fse[3];                        // 0 = literal lengths, 1 = offsets, 2 = match lengths (own tables)
    FseTable weights;                       // Huffman weight FSE table (log <= 6), reused as scratch
    int16_t norm[256 + 4];
    int16_t next[256 + 4];
    uint8_t hw[256 + 4];                    // Huffman weights
    int32_t ranks[16];
};

struct Shared : TableShared {
    uint64_t seq[SEQ_RING];                 // litLen (17: <= 131071) | matchLen (18: <= 131074) << 17 | offset (29: <= 0x1FFFFFFC) << 35
    int32_t bS[65];                         // batch execution: exclusive prefix of (litLen + matchLen), bS[n..64] = span
    int32_t bLL[64];                        // literal length per sequence
    int32_t bLP[64];                        // exclusive prefix of literal lengths
    int32_t bOF[64];                        // offset per sequence
    __attribute__((aligned(16))) uint8_t rings[2048 + 4096];  // input ring (literal / raw-block stream) + output history ring
};

struct Ctx {
    const uint8_t* __restrict__ in;
    int32_t inLen;
    uint8_t* out;
    int32_t outCap;
    uint8_t* lit;  // this wave's literal slab
    ZRings* R;     // all output bytes go through this ring pair
    int lane;
    int32_t detail;  // 0 = ok
    int32_t errOff;
};

__device__ __forceinline__ uint64_t rd_le(const Ctx& c, int32_t pos, int n)
{
    // bounds-guarded little-endian read of n <= 8 bytes; bytes outside the input read as 0
    if (pos >= 0 && pos + 8 <= c.inLen) {
        const uint64_t v = ld8(c.in + pos);
        return n >= 8 ? v : (v & ((1ull << (8 * n)) - 1ull));
    }
    uint64_t v = 0;
    for (int i = 0; i < n; i++) {
        const int32_t p = pos + i;
        if (p >= 0 && p < c.inLen) {
            v |= (uint64_t)c.in[p] << (8 * i);
        }
    }
    return v;
}

#define ZFAIL(c, d, off)         \
    {                            \
        (c).detail = (d);        \
        (c).errOff = (int32_t)(off); \
        return -1;               \
    }
#define ZVERIFY(c, cond, d, off) \
    if (!(cond)) ZFAIL(c, d, off)

__device__ __forceinline__ int32_t highest_bit(uint32_t v) { return 31 - __builtin_clz(v); }

// ---- BitInputStream.java ----
struct Bits {
    int32_t start, current;
    uint64_t bits;
    int32_t consumed;
    bool overflow;
};
__device__ __forceinline__ uint64_t peek_bits(int32_t consumed, uint64_t bits, int32_t n)  // :64-67
{
    return ((bits << (consumed & 63)) >> 1) >> ((63 - n) & 63);
}
__device__ __forceinline__ uint64_t peek_bits_fast(int32_t consumed, uint64_t bits, int32_t n)  // :74-77
{
    return (bits << (consumed & 63)) >> ((64 - n) & 63);
}
// Initializer.initialize :110-130 ; returns detail (0 = ok) and the offset through *eo
__device__ __forceinline__ int32_t bit_init(const Ctx& c, Bits& b, int32_t start, int32_t end, int32_t* eo)
{
    if (end - start < 1) {
        *eo = start;
        return ACHIP_D_ZSTD_BITSTREAM_EMPTY;
    }
    const int32_t last = (int32_t)rd_le(c, end - 1, 1);
    if (last == 0) {
        *eo = end;
        return ACHIP_D_ZSTD_BITSTREAM_NO_MARK;
    }
    b.start = start;
    b.overflow = false;
    b.consumed = 8 - highest_bit((uint32_t)last);
    const int32_t size = end - start;
    if (size >= 8) {
        b.current = end - 8;
        b.bits = rd_le(c, b.current, 8);
    }
    else {
        b.current = start;
        b.bits = rd_le(c, start, size);
        b.consumed += (8 - size) * 8;
    }
    return 0;
}
// Loader.load :171-204 ; returns the Java boolean
__device__ __forceinline__ bool bit_load(const Ctx& c, Bits& b)
{
    if (b.consumed > 64) {
        b.overflow = true;
        return true;
    }
    if (b.current == b.start) {
        return true;
    }
    int32_t bytes = (int32_t)((uint32_t)b.consumed >> 3);
    if (b.current >= b.start + 8) {
        if (bytes > 0) {
            b.current -= bytes;
            b.bits = rd_le(c, b.current, 8);
        }
        b.consumed &= 7;
    }
    else if (b.current - bytes < b.start) {
        bytes = b.current - b.start;
        b.current = b.start;
        b.consumed -= bytes * 8;
        b.bits = rd_le(c, b.start, 8);
        return true;
    }
    else {
        b.current -= bytes;
        b.consumed -= bytes * 8;
        b.bits = rd_le(c, b.current, 8);
    }
    return false;
}

// ---- FSE decoding table from normalized counts: FseTableReader.java:127-159 + spreadSymbols ----
// Wave-uniform serial code: every lane computes the same values; lane 0's LDS stores are the ones that count.  Since round 4 this is the
// FALLBACK of fse_build below (count sets whose walk does not visit every position exactly once: corrupt tables, which it reports as the Java code does).
static __device__ int32_t fse_build_serial(Ctx& c, TableShared& sh, FseTable& t, int32_t maxSymbol, int32_t tableLog, int32_t off)
{
    const int32_t tableSize = 1 << tableLog;
    int32_t high = tableSize - 1;
    __syncthreads();
    if (c.lane == 0) {
        for (int32_t s = 0; s <= maxSymbol; s++) {
            if (sh.norm[s] == -1) {
                t.e[high--] = (uint32_t)s << 16;
                sh.next[s] = 1;
            }
            else {
                sh.next[s] = sh.norm[s];
            }
        }
    }
    else {
        for (int32_t s = 0; s <= maxSymbol; s++) {
            if (sh.norm[s] == -1) {
                high--;
            }
        }
    }
    const int32_t mask = tableSize - 1;
    const int32_t step = (tableSize >> 1) + (tableSize >> 3) + 3;
    int32_t position = 0;
    for (int32_t s = 0; s <= maxSymbol; s++) {
        const int32_t n = sh.norm[s];
        for (int32_t i = 0; i < n; i++) {
            if (c.lane == 0) {
                t.e[position] = (uint32_t)s << 16;
            }
            do {
                position = (position + step) & mask;
            } while (position > high);
        }
    }
    ZVERIFY(c, position == 0, ACHIP_D_ZSTD_CORRUPTED, off);
    __syncthreads();
    if (c.lane == 0) {
        for (int32_t i = 0; i < tableSize; i++) {
            const uint32_t symbol = t.e[i] >> 16;
            const int32_t nextState = (uint16_t)sh.next[symbol]++;
            const int32_t nb = tableLog - highest_bit((uint32_t)nextState);
            const int32_t newState = (int16_t)((nextState << nb) - tableSize);
            t.e[i] = ((uint32_t)newState & 0xFFFFu) | (symbol << 16) | ((uint32_t)nb << 24);
        }
    }
    __syncthreads();
    return 0;
}


// The same table built by the wavefront (round 4; the pipeline's parse stage spent most of its time in the serial walk above: 512 states x 3
// tables per item by one lane).  What the Java code computes, restated as closed forms:
//   * symbols with count -1 ("less than one") take the table's top positions, in symbol order, one state each (FseTableReader.java:133-141);
//   * the walk  position = (position + step) & mask  visits position u as its j(u)-th stop, j(u) = u * step^-1 mod size (step is odd), and skips
//     the stops on the top positions: u is the k-th position FILLED, k = j(u) - #{top positions v with j(v) < j(u)};
//   * the symbols are handed out in symbol order, count[s] positions each: the k-th filled position gets the symbol whose running total covers k;
//   * nextState numbers a symbol's states in POSITION order from count[s] on (:143-158): rank = how many lower positions hold the same symbol.
// A lane takes positions u = lane, lane + 64, ...; ranks come from ballots over 64 positions at a time plus a running count per symbol.
// Exact for every count set whose walk fills each position once (sum of the positive counts = positions below the top ones: every table
// readFseTable accepts, the predefined ones, the Huffman weights'); anything else goes to the serial walk.
static __device__ int32_t fse_build(Ctx& c, TableShared& sh, FseTable& t, int32_t maxSymbol, int32_t tableLog, int32_t off)
{
    const int lane = c.lane;
    const int32_t tableSize = 1 << tableLog, mask = tableSize - 1;
    int16_t* const cumStart = sh.next;  // [s]: filled positions handed out before symbol s (then reused as the running counts of step C)
    __syncthreads();                    // (sh.norm is complete)
    // ---- A: lane = symbol: the top positions of the "less than one" symbols, running totals of the others ----
    int32_t lowSeen = 0, placed = 0;
    for (int32_t sBase = 0; sBase <= maxSymbol; sBase += 64) {  // (uniform)
        const int32_t sym = sBase + lane;
        const int32_t n = sym <= maxSymbol ? (int32_t)sh.norm[sym] : 0;
        const bool isLow = n == -1;
        const int32_t pos = n > 0 ? n : 0;
        const unsigned long long lowMask = __ballot(isLow);
        if (isLow) {
            const int32_t at = tableSize - 1 - (lowSeen + (int32_t)__popcll(lowMask & ((1ull << lane) - 1)));
            if (at >= 0) {
                t.e[at] = (uint32_t)sym << 16;
            }
        }
        lowSeen += (int32_t)__popcll(lowMask);
        int32_t incl = pos;
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t up = __shfl_up(incl, d);
            if (lane >= d) {
                incl += up;
            }
        }
        if (sym <= maxSymbol) {
            cumStart[sym] = (int16_t)(placed + incl - pos);
        }
        placed += __shfl(incl, 63);
    }
    const int32_t high = tableSize - 1 - lowSeen;
    if (placed != high + 1 || lowSeen > tableSize) {  // (uniform) the walk would not fill every position once: the Java loop decides
        return fse_build_serial(c, sh, t, maxSymbol, tableLog, off);
    }
    __syncthreads();
    // ---- B: lane = position: the symbol of every position below the top ones ----
    const uint32_t step = (uint32_t)((tableSize >> 1) + (tableSize >> 3) + 3);
    uint32_t inv = step;  // step^-1 mod 2^32 (Newton: the correct bits double per round, starting from 3: x * x = 1 mod 8 for odd x)
    inv *= 2u - step * inv;
    inv *= 2u - step * inv;
    inv *= 2u - step * inv;
    inv *= 2u - step * inv;
    for (int32_t u = lane; u <= high; u += 64) {
        const int32_t j = (int32_t)(((uint32_t)u * inv) & (uint32_t)mask);
        int32_t k = j;
        for (int32_t v = high + 1; v < tableSize; v++) {  // (uniform bounds) stops on top positions before this one
            k -= (int32_t)(((uint32_t)v * inv) & (uint32_t)mask) < j ? 1 : 0;
        }
        // the last symbol whose running total is <= k (symbols without positions share their successor's total: the last one wins)
        int32_t lo = 0, hi = maxSymbol;
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if ((int32_t)cumStart[mid] <= k) {
                lo = mid;
            }
            else {
                hi = mid - 1;
            }
        }
        t.e[u] = (uint32_t)lo << 16;
    }
    __syncthreads();
    // ---- C: nextState in position order: rank among the lower positions with the same symbol ----
    for (int32_t sBase = 0; sBase <= maxSymbol; sBase += 64) {
        if (sBase + lane <= maxSymbol) {
            cumStart[sBase + lane] = 0;  // (now: positions of the symbol seen so far)
        }
    }
    __syncthreads();
    for (int32_t uBase = 0; uBase < tableSize; uBase += 64) {  // (uniform)
        const int32_t u = uBase + lane;
        const bool valid = u < tableSize;
        const uint32_t symbol = valid ? (t.e[u] >> 16) : (0x100u + (uint32_t)lane);  // (lanes without a position match nobody)
        unsigned long long same = ~0ull;
#pragma unroll
        for (int bit = 0; bit < 9; bit++) {
            const bool one = ((symbol >> bit) & 1u) != 0;
            const unsigned long long m = __ballot(one);
            same &= one ? m : ~m;
        }
        const int32_t before = (int32_t)__popcll(same & ((1ull << lane) - 1));
        if (valid) {
            const int32_t n = (int32_t)sh.norm[symbol];
            const int32_t seen = (int32_t)cumStart[symbol];
            const int32_t nextState = (n == -1 ? 1 : n) + seen + before;
            const int32_t nb = tableLog - highest_bit((uint32_t)(nextState | 1));
            const int32_t newState = (int16_t)((nextState << nb) - tableSize);
            t.e[u] = ((uint32_t)newState & 0xFFFFu) | (symbol << 16) | ((uint32_t)nb << 24);
        }
        __syncthreads();  // (every lane has read the running counts)
        if (valid && before == 0) {
            cumStart[symbol] = (int16_t)((int32_t)cumStart[symbol] + (int32_t)__popcll(same));
        }
        __syncthreads();
    }
    return 0;
}

// FseTableReader.readFseTable :27-160 ; returns bytes consumed (>= 0) or -1; *logOut = table log
static __device__ int32_t read_fse_table(Ctx& c, TableShared& sh, FseTable& t, int32_t inputAddress, int32_t inputLimit, int32_t maxSymbol, int32_t maxTableLog, int32_t* logOut)
{
    int32_t input = inputAddress;
    ZVERIFY(c, inputLimit - inputAddress >= 4, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    int32_t symbolNumber = 0;
    bool previousIsZero = false;
    uint32_t bitStream = (uint32_t)rd_le(c, input, 4);
    const int32_t tableLog = (int32_t)(bitStream & 0xF) + 5;
    int32_t numberOfBits = tableLog + 1;
    bitStream >>= 4;
    int32_t bitCount = 4;
    ZVERIFY(c, tableLog <= maxTableLog, ACHIP_D_ZSTD_FSE_TABLE_LOG, input);
    int32_t remaining = (1 << tableLog) + 1;
    int32_t threshold = 1 << tableLog;
    __syncthreads();
    while (remaining > 1 && symbolNumber <= maxSymbol) {
        if (previousIsZero) {
            int32_t n0 = symbolNumber;
            while ((bitStream & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (input < inputLimit - 5) {
                    input += 2;
                    bitStream = (uint32_t)rd_le(c, input, 4) >> (bitCount & 31);
                }
                else {
                    bitStream >>= 16;
                    bitCount += 16;
                }
            }
            while ((bitStream & 3) == 3) {
                n0 += 3;
                bitStream >>= 2;
                bitCount += 2;
            }
            n0 += (int32_t)(bitStream & 3);
            bitCount += 2;
            ZVERIFY(c, n0 <= maxSymbol, ACHIP_D_ZSTD_FSE_SYMBOL, input);
            while (symbolNumber < n0) {
                if (c.lane == 0) sh.norm[symbolNumber] = 0;
                symbolNumber++;
            }
            if ((input <= inputLimit - 7) || (input + (bitCount >> 3) <= inputLimit - 4)) {
                input += bitCount >> 3;
                bitCount &= 7;
                bitStream = (uint32_t)rd_le(c, input, 4) >> (bitCount & 31);
            }
            else {
                bitStream >>= 2;
            }
        }
        const int16_t max = (int16_t)((2 * threshold - 1) - remaining);
        int16_t count;
        if ((int32_t)(bitStream & (uint32_t)(threshold - 1)) < max) {
            count = (int16_t)(bitStream & (uint32_t)(threshold - 1));
            bitCount += numberOfBits - 1;
        }
        else {
            count = (int16_t)(bitStream & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) {
                count = (int16_t)(count - max);
            }
            bitCount += numberOfBits;
        }
        count--;
        remaining -= count < 0 ? -count : count;
        if (c.lane == 0) sh.norm[symbolNumber] = count;
        symbolNumber++;
        previousIsZero = count == 0;
        while (remaining < threshold) {
            numberOfBits--;
            threshold >>= 1;
        }
        if ((input <= inputLimit - 7) || (input + (bitCount >> 3) <= inputLimit - 4)) {
            input += bitCount >> 3;
            bitCount &= 7;
        }
        else {
            bitCount -= 8 * (inputLimit - 4 - input);
            input = inputLimit - 4;
        }
        bitStream = (uint32_t)rd_le(c, input, 4) >> (bitCount & 31);
    }
    ZVERIFY(c, remaining == 1 && bitCount <= 32, ACHIP_D_ZSTD_CORRUPTED, input);
    maxSymbol = symbolNumber - 1;
    ZVERIFY(c, maxSymbol <= 255, ACHIP_D_ZSTD_FSE_SYMBOL, input);
    input += (bitCount + 7) >> 3;
    if (fse_build(c, sh, t, maxSymbol, tableLog, input) < 0) {
        return -1;
    }
    *logOut = tableLog;
    return input - inputAddress;
}

#define FSE_NEWSTATE(e) ((int32_t)(int16_t)((e) & 0xFFFFu))
#define FSE_SYMBOL(e) ((int32_t)(((e) >> 16) & 0xFFu))
#define FSE_NBITS(e) ((int32_t)((e) >> 24))

// FiniteStateEntropy.decompress :38-151 (Huffman weights) into sh.hw ; returns count or -1
static __device__ int32_t fse_decompress_weights(Ctx& c, TableShared& sh, const FseTable& t, int32_t log, int32_t inputAddress, int32_t inputLimit)
{
    const int32_t outputLimit = 256;
    int32_t output = 0;
    Bits b;
    int32_t eo = 0;
    const int32_t d = bit_init(c, b, inputAddress, inputLimit, &eo);
    if (d != 0) ZFAIL(c, d, eo);
    int32_t state1 = (int32_t)peek_bits(b.consumed, b.bits, log);
    b.consumed += log;
    bit_load(c, b);
    int32_t state2 = (int32_t)peek_bits(b.consumed, b.bits, log);
    b.consumed += log;
    bit_load(c, b);
#define W_EMIT(state)                                   \
    {                                                   \
        if (c.lane == 0) sh.hw[output] = (uint8_t)FSE_SYMBOL(t.e[state]); \
        output++;                                       \
    }
#define W_STEP(state)                                                                      \
    {                                                                                      \
        const uint32_t e_ = t.e[state];                                                    \
        const int32_t nb_ = FSE_NBITS(e_);                                                 \
        state = FSE_NEWSTATE(e_) + (int32_t)peek_bits(b.consumed, b.bits, nb_);            \
        b.consumed += nb_;                                                                 \
    }
    while (output <= outputLimit - 4) {
        W_EMIT(state1) W_STEP(state1) W_EMIT(state2) W_STEP(state2) W_EMIT(state1) W_STEP(state1) W_EMIT(state2) W_STEP(state2)
        if (bit_load(c, b)) {
            break;
        }
    }
    for (;;) {
        ZVERIFY(c, output <= outputLimit - 2, ACHIP_D_ZSTD_FSE_OUTPUT_SMALL, inputAddress);
        W_EMIT(state1) W_STEP(state1)
        b.overflow = false;
        bit_load(c, b);
        if (b.overflow) {
            W_EMIT(state2)
            break;
        }
        ZVERIFY(c, output <= outputLimit - 2, ACHIP_D_ZSTD_FSE_OUTPUT_SMALL, inputAddress);
        W_EMIT(state2) W_STEP(state2)
        b.overflow = false;
        bit_load(c, b);
        if (b.overflow) {
            W_EMIT(state1)
            break;
        }
    }
#undef W_EMIT
#undef W_STEP
    __syncthreads();
    return output;
}

// Huffman.readTable :52-128 ; returns bytes consumed or -1 ; sets *tableLogOut
static __device__ int32_t huf_read_table(Ctx& c, TableShared& sh, int32_t inputAddress, int32_t size, int32_t* tableLogOut)
{
    int32_t input = inputAddress;
    ZVERIFY(c, size > 0, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    int32_t inputSize = (int32_t)rd_le(c, input++, 1);
    int32_t outputSize;
    __syncthreads();
    if (inputSize >= 128) {
        outputSize = inputSize - 127;
        inputSize = (outputSize + 1) / 2;
        ZVERIFY(c, inputSize + 1 <= size, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        ZVERIFY(c, outputSize <= 256, ACHIP_D_ZSTD_CORRUPTED, input);
        for (int32_t i = c.lane * 2; i < outputSize; i += 128) {
            const int32_t value = (int32_t)rd_le(c, input + i / 2, 1);
            sh.hw[i] = (uint8_t)(value >> 4);
            sh.hw[i + 1] = (uint8_t)(value & 0xF);
        }
        __syncthreads();
    }
    else {
        ZVERIFY(c, inputSize + 1 <= size, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        const int32_t inputLimit = input + inputSize;
        int32_t wlog = 0;
        const int32_t n = read_fse_table(c, sh, sh.weights, input, inputLimit, 255, 6, &wlog);
        if (n < 0) return -1;
        input += n;
        outputSize = fse_decompress_weights(c, sh, sh.weights, wlog, input, inputLimit);
        if (outputSize < 0) return -1;
    }

    // rank statistics, a lane per symbol (round 4; was a serial loop over the <= 256 weights): how many symbols have each weight, the total weight
    int32_t totalWeight = 0;
    int32_t ranks[HUF_MAX_TABLE_LOG + 1];
#pragma unroll
    for (int i = 0; i <= HUF_MAX_TABLE_LOG; i++) ranks[i] = 0;
    for (int32_t sBase = 0; sBase < outputSize; sBase += 64) {  // (uniform)
        const int32_t w = sBase + c.lane < outputSize ? (int32_t)sh.hw[sBase + c.lane] : -1;
        ZVERIFY(c, __ballot(w > HUF_MAX_TABLE_LOG) == 0, ACHIP_D_ZSTD_CORRUPTED, input);  // Java: ArrayIndexOutOfBoundsException
#pragma unroll
        for (int k = 0; k <= HUF_MAX_TABLE_LOG; k++) {
            const int32_t n = (int32_t)__popcll(__ballot(w == k));
            ranks[k] += n;
            totalWeight += n * ((1 << k) >> 1);
        }
    }
    ZVERIFY(c, totalWeight != 0, ACHIP_D_ZSTD_CORRUPTED, input);
    const int32_t tableLog = highest_bit((uint32_t)totalWeight) + 1;
    ZVERIFY(c, tableLog <= HUF_MAX_TABLE_LOG, ACHIP_D_ZSTD_CORRUPTED, input);
    const int32_t total = 1 << tableLog;
    const int32_t rest = total - totalWeight;
    ZVERIFY(c, (rest & (rest - 1)) == 0, ACHIP_D_ZSTD_CORRUPTED, input);
    const int32_t lastWeight = highest_bit((uint32_t)rest) + 1;
    ZVERIFY(c, outputSize <= 255, ACHIP_D_ZSTD_CORRUPTED, input);  // Java: weights[256] out of bounds
    __syncthreads();
    if (c.lane == 0) {
        sh.hw[outputSize] = (uint8_t)lastWeight;
    }
#pragma unroll
    for (int k = 0; k <= HUF_MAX_TABLE_LOG; k++) ranks[k] += (lastWeight == k);
    const int32_t numberOfSymbols = outputSize + 1;

    // where each weight's symbols start (:100-108); ranks[] turns from counts into starts, r1 = where the weight-1 symbols end
    int32_t nextRankStart = 0;
    const int32_t count1 = ranks[1];
#pragma unroll
    for (int i = 1; i <= HUF_MAX_TABLE_LOG; i++) {
        if (i < tableLog + 1) {
            const int32_t current = nextRankStart;
            nextRankStart += ranks[i] << (i - 1);
            ranks[i] = current;
        }
    }
    const int32_t r1 = ranks[1] + count1;
    __syncthreads();
    // populate (:110-123): symbol n occupies [start(n), start(n) + length), start = its weight's start + (symbols of equal weight before n) * length.
    // A lane per symbol finds the starts (ballots per weight + the counts of the chunks before); then the wavefront fills symbol after symbol,
    // 64 entries per step (a weight-11 symbol is 1024 entries; lane 0 alone used to write all 2048 of the table).
    int16_t* const symStart = sh.next;
    int32_t seenOfWeight[HUF_MAX_TABLE_LOG + 1];
#pragma unroll
    for (int i = 0; i <= HUF_MAX_TABLE_LOG; i++) seenOfWeight[i] = 0;
    for (int32_t sBase = 0; sBase < numberOfSymbols; sBase += 64) {  // (uniform)
        const int32_t n = sBase + c.lane;
        const int32_t w = n < numberOfSymbols ? (int32_t)sh.hw[n] : -1;
        int32_t start = 0;
#pragma unroll
        for (int k = 1; k <= HUF_MAX_TABLE_LOG; k++) {
            const unsigned long long m = __ballot(w == k);
            if (w == k) {
                start = ranks[k] + ((seenOfWeight[k] + (int32_t)__popcll(m & ((1ull << c.lane) - 1))) << (k - 1));
            }
            seenOfWeight[k] += (int32_t)__popcll(m);
        }
        if (n < numberOfSymbols) {
            symStart[n] = (int16_t)start;
        }
    }
    __syncthreads();
    for (int32_t n = 0; n < numberOfSymbols; n++) {  // (uniform)
        const int32_t weight = sh.hw[n];
        const int32_t length = (1 << weight) >> 1;
        const uint16_t entry = (uint16_t)(n | ((tableLog + 1 - weight) << 8));
        const int32_t begin = symStart[n];
        for (int32_t i = c.lane; i < length; i += 64) {
            sh.huf[begin + i] = entry;
        }
    }
    __syncthreads();
    ZVERIFY(c, r1 >= 2 && (r1 & 1) == 0, ACHIP_D_ZSTD_CORRUPTED, input);
    *tableLogOut = tableLog;
    return inputSize + 1;
}

// Huffman.decodeSymbol :319-324
__device__ __forceinline__ int32_t huf_symbol(const uint16_t* huf, int32_t tableLog, uint64_t bits, int32_t& consumed)
{
    const uint32_t e = huf[(int32_t)peek_bits_fast(consumed, bits, tableLog)];
    consumed += (int32_t)(e >> 8);
    return (int32_t)(e & 0xFF);
}

// One Huffman stream (decodeSingleStream :130-164 body + decodeTail :291-317) decoded by the calling lane.
// Returns 0 or ACHIP_D_ZSTD_BITSTREAM_NOT_CONSUMED.
__device__ __forceinline__ int32_t huf_decode_stream(const Ctx& c, const uint16_t* huf, int32_t tableLog, Bits& b, uint8_t* out, int32_t output, int32_t outputLimit)
{
    const int32_t fastLimit = outputLimit - 4;
    bool done = false;
    while (output < fastLimit) {
        if (bit_load(c, b)) {
            done = true;
            break;
        }
        uint32_t w = (uint32_t)huf_symbol(huf, tableLog, b.bits, b.consumed);
        w |= (uint32_t)huf_symbol(huf, tableLog, b.bits, b.consumed) << 8;
        w |= (uint32_t)huf_symbol(huf, tableLog, b.bits, b.consumed) << 16;
        w |= (uint32_t)huf_symbol(huf, tableLog, b.bits, b.consumed) << 24;
        st4(out + output, w);
        output += 4;
    }
    if (!done) {
        while (output < outputLimit) {
            if (bit_load(c, b)) {
                break;
            }
            out[output++] = (uint8_t)huf_symbol(huf, tableLog, b.bits, b.consumed);
        }
    }
    while (output < outputLimit) {
        out[output++] = (uint8_t)huf_symbol(huf, tableLog, b.bits, b.consumed);
    }
    return (b.start == b.current && b.consumed == 64) ? 0 : ACHIP_D_ZSTD_BITSTREAM_NOT_CONSUMED;
}

// wave copy with byte-exact bounds (src, dst do not overlap)
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, int32_t n, int lane) { group_copy<64>(dst, src, n, lane); }
}  // namespace zd
}  // namespace achip
