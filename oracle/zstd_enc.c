/*
 * oracle/zstd_enc.c -- CPU restatement of the reference's Java Zstd encoder (always level 3 = DFAST).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   frame / block assembly ... M/zstd/ZstdFrameCompressor.java:52-432
 *   parameters ............... M/zstd/CompressionParameters.java:36-145,256-324
 *   match finder ............. M/zstd/DoubleFastBlockCompressor.java:28-256, BlockCompressionState.java, RepeatedOffsets.java
 *   sequences ................ M/zstd/SequenceStore.java:20-160, SequenceEncoder.java:34-342
 *   FSE ...................... M/zstd/FseCompressionTable.java:18-155, FiniteStateEntropy.java:153-521
 *   Huffman .................. M/zstd/HuffmanCompressionTable.java:27-437, HuffmanCompressor.java:26-135,
 *                              HuffmanCompressionContext.java, NodeTable.java, workspaces
 *   bit output ............... M/zstd/BitOutputStream.java:20-90, Histogram.java:21-65
 *
 * Java semantics kept: int/short/byte narrowing where the Java code narrows, 6-bit masked long shifts,
 * the 8-byte putLong of BitOutputStream.flush (so bytes past the stream end are disturbed exactly as in Java,
 * and later overwritten), array lifetimes of the per-frame CompressionContext.
 * Java `checkArgument` failures ("Output buffer too small") map to ACHIP_CLASS_OUTPUT_TOO_SMALL; the two
 * unchecked-exception corners of HuffmanCompressionTable.write (:240-244, :262 with maxSymbol == 255) map to
 * ACHIP_CLASS_INVALID_ARGUMENT / ACHIP_D_UNSUPPORTED.
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

#define SIZE_OF_LONG 8
#define MAGIC_NUMBER 0xFD2FB528u
#define MIN_WINDOW_LOG 10
#define MAX_WINDOW_LOG 31
#define SIZE_OF_BLOCK_HEADER 3
#define MIN_BLOCK_SIZE 3
#define MAX_BLOCK_SIZE (128 * 1024)
#define RAW_BLOCK 0
#define COMPRESSED_BLOCK 2
#define RAW_LITERALS_BLOCK 0
#define RLE_LITERALS_BLOCK 1
#define COMPRESSED_LITERALS_BLOCK 2
#define TREELESS_LITERALS_BLOCK 3
#define SEQUENCE_ENCODING_BASIC 0
#define SEQUENCE_ENCODING_RLE 1
#define SEQUENCE_ENCODING_COMPRESSED 2
#define MAX_LITERALS_LENGTH_SYMBOL 35
#define MAX_MATCH_LENGTH_SYMBOL 52
#define MAX_OFFSET_CODE_SYMBOL 31
#define DEFAULT_MAX_OFFSET_CODE_SYMBOL 28
#define LITERAL_LENGTH_TABLE_LOG 9
#define MATCH_LENGTH_TABLE_LOG 9
#define OFFSET_TABLE_LOG 8
#define LONG_NUMBER_OF_SEQUENCES 0x7F00
#define HUF_MAX_SYMBOL 255
#define HUF_MAX_SYMBOL_COUNT 256
#define HUF_MAX_TABLE_LOG 12
#define HUF_MIN_TABLE_LOG 5
#define HUF_MAX_FSE_TABLE_LOG 6
#define FSE_MAX_SYMBOL 255
#define FSE_MAX_TABLE_LOG 12
#define FSE_MIN_TABLE_LOG 5
#define MAX_FRAME_HEADER_SIZE 14
#define MINIMUM_LITERALS_SIZE 63
#define MAX_HUFFMAN_TABLE_LOG 11

static const uint8_t LITERALS_LENGTH_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint8_t MATCH_LENGTH_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                              1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

typedef struct {
    jmp_buf jb;
    int32_t status;
} fail_ctx;
static __thread fail_ctx* g_fail;

static void fail(int cls, int detail)
{
    g_fail->status = ACHIP_STATUS(cls, detail);
    longjmp(g_fail->jb, 1);
}
#define CHECK_ARGUMENT(cond) \
    do {                     \
        if (!(cond)) fail(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_ZSTD_MAX_OUTPUT); \
    } while (0)

static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void st64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
static inline void st32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
static inline void st16(uint8_t* p, uint32_t v) { uint16_t w = (uint16_t)v; memcpy(p, &w, 2); }
static inline void st24(uint8_t* p, uint32_t v) { st16(p, v); p[2] = (uint8_t)(v >> 16); } /* Util.put24BitLittleEndian */
static inline int32_t highest_bit(uint32_t v) { return 31 - __builtin_clz(v); }           /* Util.highestBit */

/* Util.minTableLog :139-149 */
static int32_t min_table_log(int32_t inputSize, int32_t maxSymbolValue)
{
    int32_t minBitsSrc = highest_bit((uint32_t)(inputSize - 1)) + 1;
    int32_t minBitsSymbols = highest_bit((uint32_t)maxSymbolValue) + 2;
    return minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols;
}

/* ---- BitOutputStream.java ---- */
typedef struct {
    uint8_t* base;       /* output array (absolute indexing below) */
    int64_t outputAddress, outputLimit, currentAddress;
    uint64_t container;
    int32_t bitCount;
} bitout;

static void bo_init(bitout* s, uint8_t* base, int64_t outputAddress, int32_t outputSize) /* :41-50 */
{
    CHECK_ARGUMENT(outputSize >= SIZE_OF_LONG);
    s->base = base;
    s->outputAddress = outputAddress;
    s->outputLimit = outputAddress + outputSize - SIZE_OF_LONG;
    s->currentAddress = outputAddress;
    s->container = 0;
    s->bitCount = 0;
}
static inline void bo_add_bits(bitout* s, int32_t value, int32_t bits) /* :52-56 (BIT_MASK: up to 31 bits) */
{
    const uint64_t mask = bits >= 32 ? 0xFFFFFFFFull : ((1ull << bits) - 1);
    s->container |= ((uint64_t)(int64_t)value & mask) << (s->bitCount & 63);
    s->bitCount += bits;
}
static inline void bo_add_bits_fast(bitout* s, int32_t value, int32_t bits) /* :61-65 */
{
    s->container |= (uint64_t)(int64_t)value << (s->bitCount & 63);
    s->bitCount += bits;
}
static inline void bo_flush(bitout* s) /* :67-80 */
{
    int32_t bytes = (int32_t)((uint32_t)s->bitCount >> 3);
    st64(s->base + s->currentAddress, s->container);
    s->currentAddress += bytes;
    if (s->currentAddress > s->outputLimit) {
        s->currentAddress = s->outputLimit;
    }
    s->bitCount &= 7;
    s->container >>= ((bytes * 8) & 63);
}
static int32_t bo_close(bitout* s) /* :82-92 */
{
    bo_add_bits_fast(s, 1, 1);
    bo_flush(s);
    if (s->currentAddress >= s->outputLimit) {
        return 0;
    }
    return (int32_t)((s->currentAddress - s->outputAddress) + (s->bitCount > 0 ? 1 : 0));
}

/* ---- FseCompressionTable.java ---- */
typedef struct {
    int16_t nextState[1 << 9];
    int32_t deltaNumberOfBits[FSE_MAX_SYMBOL + 1];
    int32_t deltaFindState[FSE_MAX_SYMBOL + 1];
    int32_t log2Size;
} fse_ctable;

static void fse_init_rle(fse_ctable* t, int32_t symbol) /* :46-55 */
{
    t->log2Size = 0;
    t->nextState[0] = 0;
    t->nextState[1] = 0;
    t->deltaFindState[symbol] = 0;
    t->deltaNumberOfBits[symbol] = 0;
}

static void fse_initialize(fse_ctable* t, const int16_t* norm, int32_t maxSymbol, int32_t tableLog) /* :57-117 */
{
    int32_t tableSize = 1 << tableLog;
    uint8_t table[1 << 9];
    int32_t highThreshold = tableSize - 1;
    int32_t cumulative[FSE_MAX_SYMBOL + 2];
    t->log2Size = tableLog;
    cumulative[0] = 0;
    for (int32_t i = 1; i <= maxSymbol + 1; i++) {
        if (norm[i - 1] == -1) {
            cumulative[i] = cumulative[i - 1] + 1;
            table[highThreshold--] = (uint8_t)(i - 1);
        }
        else {
            cumulative[i] = cumulative[i - 1] + norm[i - 1];
        }
    }
    cumulative[maxSymbol + 1] = tableSize + 1;

    /* spreadSymbols :138-154 */
    int32_t mask = tableSize - 1;
    int32_t step = (tableSize >> 1) + (tableSize >> 3) + 3;
    int32_t position = 0;
    for (int32_t symbol = 0; symbol <= maxSymbol; symbol++) {
        for (int32_t i = 0; i < norm[symbol]; i++) {
            table[position] = (uint8_t)symbol;
            do {
                position = (position + step) & mask;
            }
            while (position > highThreshold);
        }
    }
    /* position != 0 => AssertionError in Java; cannot happen for normalised counts summing to tableSize */

    for (int32_t i = 0; i < tableSize; i++) {
        uint8_t symbol = table[i];
        t->nextState[cumulative[symbol]++] = (int16_t)(tableSize + i);
    }

    int32_t total = 0;
    for (int32_t symbol = 0; symbol <= maxSymbol; symbol++) {
        int32_t n = norm[symbol];
        if (n == 0) {
            t->deltaNumberOfBits[symbol] = ((tableLog + 1) << 16) - tableSize;
        }
        else if (n == -1 || n == 1) {
            t->deltaNumberOfBits[symbol] = (tableLog << 16) - tableSize;
            t->deltaFindState[symbol] = total - 1;
            total++;
        }
        else {
            int32_t maxBitsOut = tableLog - highest_bit((uint32_t)(n - 1));
            int32_t minStatePlus = n << maxBitsOut;
            t->deltaNumberOfBits[symbol] = (maxBitsOut << 16) - minStatePlus;
            t->deltaFindState[symbol] = total - n;
            total += n;
        }
    }
}

static inline int32_t fse_begin(const fse_ctable* t, int32_t symbol) /* :119-124 */
{
    int32_t outputBits = (int32_t)((uint32_t)(t->deltaNumberOfBits[symbol] + (1 << 15)) >> 16);
    int32_t base = (int32_t)((uint32_t)((outputBits << 16) - t->deltaNumberOfBits[symbol]) >> (outputBits & 31));
    return t->nextState[base + t->deltaFindState[symbol]];
}
static inline int32_t fse_encode(const fse_ctable* t, bitout* s, int32_t state, int32_t symbol) /* :126-131 */
{
    int32_t outputBits = (int32_t)((uint32_t)(state + t->deltaNumberOfBits[symbol]) >> 16);
    bo_add_bits(s, state, outputBits);
    return t->nextState[(int32_t)((uint32_t)state >> (outputBits & 31)) + t->deltaFindState[symbol]];
}
static inline void fse_finish(const fse_ctable* t, bitout* s, int32_t state) /* :133-137 */
{
    bo_add_bits(s, state, t->log2Size);
    bo_flush(s);
}

/* ---- FiniteStateEntropy.java (compression side) ---- */
static int32_t fse_optimal_table_log(int32_t maxTableLog, int32_t inputSize, int32_t maxSymbol) /* :236-255 */
{
    int32_t result = maxTableLog;
    int32_t a = highest_bit((uint32_t)(inputSize - 1)) - 2;
    if (a < result) result = a;
    int32_t b = min_table_log(inputSize, maxSymbol);
    if (b > result) result = b;
    if (result < FSE_MIN_TABLE_LOG) result = FSE_MIN_TABLE_LOG;
    if (result > FSE_MAX_TABLE_LOG) result = FSE_MAX_TABLE_LOG;
    return result;
}

static const int32_t REST_TO_BEAT[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
#define UNASSIGNED (-2)

static void fse_normalize_counts2(int16_t* norm, int32_t tableLog, const int32_t* counts, int32_t total, int32_t maxSymbol) /* :318-405 */
{
    int32_t distributed = 0;
    int32_t lowThreshold = (int32_t)((uint32_t)total >> tableLog);
    int32_t lowOne = (int32_t)((uint32_t)(total * 3) >> (tableLog + 1));
    for (int32_t i = 0; i <= maxSymbol; i++) {
        if (counts[i] == 0) {
            norm[i] = 0;
        }
        else if (counts[i] <= lowThreshold) {
            norm[i] = -1;
            distributed++;
            total -= counts[i];
        }
        else if (counts[i] <= lowOne) {
            norm[i] = 1;
            distributed++;
            total -= counts[i];
        }
        else {
            norm[i] = UNASSIGNED;
        }
    }
    int32_t normalizationFactor = 1 << tableLog;
    int32_t toDistribute = normalizationFactor - distributed;
    if ((total / toDistribute) > lowOne) {
        lowOne = ((total * 3) / (toDistribute * 2));
        for (int32_t i = 0; i <= maxSymbol; i++) {
            if (norm[i] == UNASSIGNED && counts[i] <= lowOne) {
                norm[i] = 1;
                distributed++;
                total -= counts[i];
            }
        }
        toDistribute = normalizationFactor - distributed;
    }
    if (distributed == maxSymbol + 1) {
        int32_t maxValue = 0, maxCount = 0;
        for (int32_t i = 0; i <= maxSymbol; i++) {
            if (counts[i] > maxCount) {
                maxValue = i;
                maxCount = counts[i];
            }
        }
        norm[maxValue] = (int16_t)(norm[maxValue] + (int16_t)toDistribute);
        return;
    }
    if (total == 0) {
        for (int32_t i = 0; toDistribute > 0; i = (i + 1) % (maxSymbol + 1)) {
            if (norm[i] > 0) {
                toDistribute--;
                norm[i]++;
            }
        }
        return;
    }
    int64_t vStepLog = 62 - tableLog;
    int64_t mid = (1LL << (vStepLog - 1)) - 1;
    int64_t rStep = (((1LL << vStepLog) * toDistribute) + mid) / total;
    int64_t tmpTotal = mid;
    for (int32_t i = 0; i <= maxSymbol; i++) {
        if (norm[i] == UNASSIGNED) {
            int64_t end = tmpTotal + ((int64_t)counts[i] * rStep);
            int32_t sStart = (int32_t)((uint64_t)tmpTotal >> vStepLog);
            int32_t sEnd = (int32_t)((uint64_t)end >> vStepLog);
            int32_t weight = sEnd - sStart;
            norm[i] = (int16_t)weight;
            tmpTotal = end;
        }
    }
}

static void fse_normalize_counts(int16_t* norm, int32_t tableLog, const int32_t* counts, int32_t total, int32_t maxSymbol) /* :257-316 */
{
    int64_t scale = 62 - tableLog;
    int64_t step = (1LL << 62) / total;
    int64_t vstep = 1LL << (scale - 20);
    int32_t stillToDistribute = 1 << tableLog;
    int32_t largest = 0;
    int16_t largestProbability = 0;
    int32_t lowThreshold = (int32_t)((uint32_t)total >> tableLog);

    for (int32_t symbol = 0; symbol <= maxSymbol; symbol++) {
        if (counts[symbol] == 0) {
            norm[symbol] = 0;
            continue;
        }
        if (counts[symbol] <= lowThreshold) {
            norm[symbol] = -1;
            stillToDistribute--;
        }
        else {
            int16_t probability = (int16_t)((uint64_t)((int64_t)counts[symbol] * step) >> scale);
            if (probability < 8) {
                int64_t restToBeat = vstep * REST_TO_BEAT[probability];
                int64_t delta = (int64_t)counts[symbol] * step - (((int64_t)probability) << scale);
                if (delta > restToBeat) {
                    probability++;
                }
            }
            if (probability > largestProbability) {
                largestProbability = probability;
                largest = symbol;
            }
            norm[symbol] = probability;
            stillToDistribute -= probability;
        }
    }
    if (-stillToDistribute >= (int32_t)((uint32_t)(int32_t)norm[largest] >> 1)) {
        fse_normalize_counts2(norm, tableLog, counts, total, maxSymbol);
    }
    else {
        norm[largest] = (int16_t)(norm[largest] + (int16_t)stillToDistribute);
    }
}

static int32_t fse_write_normalized_counts(uint8_t* base, int64_t outputAddress, int32_t outputSize, const int16_t* norm, int32_t maxSymbol, int32_t tableLog) /* :407-521 */
{
    int64_t output = outputAddress;
    int64_t outputLimit = outputAddress + outputSize;
    int32_t tableSize = 1 << tableLog;
    int32_t bitCount = 0;
    int32_t bitStream = tableLog - FSE_MIN_TABLE_LOG;
    bitCount += 4;
    int32_t remaining = tableSize + 1;
    int32_t threshold = tableSize;
    int32_t tableBitCount = tableLog + 1;
    int32_t symbol = 0;
    int previousIs0 = 0;
    while (remaining > 1) {
        if (previousIs0) {
            int32_t start = symbol;
            while (norm[symbol] == 0) {
                symbol++;
            }
            while (symbol >= start + 24) {
                start += 24;
                bitStream |= (int32_t)((uint32_t)0xFFFF << (bitCount & 31));
                CHECK_ARGUMENT(output + 2 <= outputLimit);
                st16(base + output, (uint32_t)bitStream);
                output += 2;
                bitStream = (int32_t)((uint32_t)bitStream >> 16);
            }
            while (symbol >= start + 3) {
                start += 3;
                bitStream |= (int32_t)((uint32_t)3 << (bitCount & 31));
                bitCount += 2;
            }
            bitStream |= (int32_t)((uint32_t)(symbol - start) << (bitCount & 31));
            bitCount += 2;
            if (bitCount > 16) {
                CHECK_ARGUMENT(output + 2 <= outputLimit);
                st16(base + output, (uint32_t)bitStream);
                output += 2;
                bitStream = (int32_t)((uint32_t)bitStream >> 16);
                bitCount -= 16;
            }
        }
        int32_t count = norm[symbol++];
        int32_t max = (2 * threshold - 1) - remaining;
        remaining -= count < 0 ? -count : count;
        count++;
        if (count >= threshold) {
            count += max;
        }
        bitStream |= (int32_t)((uint32_t)count << (bitCount & 31));
        bitCount += tableBitCount;
        bitCount -= (count < max ? 1 : 0);
        previousIs0 = (count == 1);
        while (remaining < threshold) {
            tableBitCount--;
            threshold >>= 1;
        }
        if (bitCount > 16) {
            CHECK_ARGUMENT(output + 2 <= outputLimit);
            st16(base + output, (uint32_t)bitStream);
            output += 2;
            bitStream = (int32_t)((uint32_t)bitStream >> 16);
            bitCount -= 16;
        }
    }
    CHECK_ARGUMENT(output + 2 <= outputLimit);
    st16(base + output, (uint32_t)bitStream);
    output += (bitCount + 7) / 8;
    CHECK_ARGUMENT(symbol <= maxSymbol + 1);
    return (int32_t)(output - outputAddress);
}

/* FiniteStateEntropy.compress :158-234 (input = byte[] of Huffman weights) */
static int32_t fse_compress(uint8_t* base, int64_t outputAddress, int32_t outputSize, const uint8_t* in, int32_t inputSize, const fse_ctable* table)
{
    CHECK_ARGUMENT(outputSize >= SIZE_OF_LONG);
    int32_t input = inputSize;
    if (inputSize <= 2) {
        return 0;
    }
    bitout stream;
    bo_init(&stream, base, outputAddress, outputSize);
    int32_t state1, state2;
    if ((inputSize & 1) != 0) {
        input--;
        state1 = fse_begin(table, in[input]);
        input--;
        state2 = fse_begin(table, in[input]);
        input--;
        state1 = fse_encode(table, &stream, state1, in[input]);
        bo_flush(&stream);
    }
    else {
        input--;
        state2 = fse_begin(table, in[input]);
        input--;
        state1 = fse_begin(table, in[input]);
    }
    inputSize -= 2;
    if ((inputSize & 2) != 0) { /* SIZE_OF_LONG * 8 > MAX_TABLE_LOG * 4 + 7 is true */
        input--;
        state2 = fse_encode(table, &stream, state2, in[input]);
        input--;
        state1 = fse_encode(table, &stream, state1, in[input]);
        bo_flush(&stream);
    }
    while (input > 0) {
        input--;
        state2 = fse_encode(table, &stream, state2, in[input]);
        input--;
        state1 = fse_encode(table, &stream, state1, in[input]);
        input--;
        state2 = fse_encode(table, &stream, state2, in[input]);
        input--;
        state1 = fse_encode(table, &stream, state1, in[input]);
        bo_flush(&stream);
    }
    fse_finish(table, &stream, state2);
    fse_finish(table, &stream, state1);
    return bo_close(&stream);
}

/* ---- Histogram.java ---- */
static void histogram_count(const uint8_t* in, int32_t n, int32_t* counts, int32_t countsLen)
{
    memset(counts, 0, sizeof(int32_t) * (size_t)countsLen);
    for (int32_t i = 0; i < n; i++) {
        counts[in[i]]++;
    }
}
static int32_t find_largest_count(const int32_t* counts, int32_t maxSymbol)
{
    int32_t max = 0;
    for (int32_t i = 0; i <= maxSymbol; i++) {
        if (counts[i] > max) max = counts[i];
    }
    return max;
}
static int32_t find_max_symbol(const int32_t* counts, int32_t maxSymbol)
{
    while (counts[maxSymbol] == 0) {
        maxSymbol--;
    }
    return maxSymbol;
}

/* ---- Huffman compression tables ---- */
typedef struct {
    int16_t values[HUF_MAX_SYMBOL_COUNT];
    uint8_t numberOfBits[HUF_MAX_SYMBOL_COUNT];
    int32_t maxSymbol;
    int32_t maxNumberOfBits;
} huf_ctable;

typedef struct {
    int32_t count[2 * HUF_MAX_SYMBOL_COUNT - 1];
    int16_t parents[2 * HUF_MAX_SYMBOL_COUNT - 1];
    int32_t symbols[2 * HUF_MAX_SYMBOL_COUNT - 1];
    uint8_t numberOfBits[2 * HUF_MAX_SYMBOL_COUNT - 1];
} node_table;

typedef struct {
    /* HuffmanCompressionTableWorkspace */
    node_table nodeTable;
    int16_t entriesPerRank[HUF_MAX_TABLE_LOG + 1];
    int16_t valuesPerRank[HUF_MAX_TABLE_LOG + 1];
    int32_t rankLast[HUF_MAX_TABLE_LOG + 2];
    /* HuffmanTableWriterWorkspace */
    uint8_t weights[HUF_MAX_SYMBOL + 1]; /* Java: byte[MAX_SYMBOL]; index 255 is the :262 out-of-bounds corner */
    int32_t wcounts[HUF_MAX_TABLE_LOG + 1];
    int16_t wnorm[HUF_MAX_TABLE_LOG + 1];
    fse_ctable wfse;
    /* HuffmanCompressionContext */
    huf_ctable tables[2];
    int previousTable, temporaryTable, previousCandidate, temporaryCandidate; /* indices into tables[] */
} huf_context;

static int32_t huf_optimal_number_of_bits(int32_t maxNumberOfBits, int32_t inputSize, int32_t maxSymbol) /* :42-57 */
{
    int32_t result = maxNumberOfBits;
    int32_t a = highest_bit((uint32_t)(inputSize - 1)) - 1;
    if (a < result) result = a;
    int32_t b = min_table_log(inputSize, maxSymbol);
    if (b > result) result = b;
    if (result < HUF_MIN_TABLE_LOG) result = HUF_MIN_TABLE_LOG;
    if (result > HUF_MAX_TABLE_LOG) result = HUF_MAX_TABLE_LOG;
    return result;
}

static void node_copy(node_table* t, int32_t from, int32_t to)
{
    t->count[to] = t->count[from];
    t->parents[to] = t->parents[from];
    t->symbols[to] = t->symbols[from];
    t->numberOfBits[to] = t->numberOfBits[from];
}

static int32_t huf_build_tree(const int32_t* counts, int32_t maxSymbol, node_table* nt) /* :105-190 */
{
    int16_t current = 0;
    for (int32_t symbol = 0; symbol <= maxSymbol; symbol++) {
        int32_t count = counts[symbol];
        int32_t position = current;
        while (position > 1 && count > nt->count[position - 1]) {
            node_copy(nt, position - 1, position);
            position--;
        }
        nt->count[position] = count;
        nt->symbols[position] = symbol;
        current++;
    }
    int32_t lastNonZero = maxSymbol;
    while (nt->count[lastNonZero] == 0) {
        lastNonZero--;
    }
    int16_t nonLeafStart = HUF_MAX_SYMBOL_COUNT;
    current = nonLeafStart;
    int32_t currentLeaf = lastNonZero;
    int32_t currentNonLeaf = current;
    nt->count[current] = nt->count[currentLeaf] + nt->count[currentLeaf - 1];
    nt->parents[currentLeaf] = current;
    nt->parents[currentLeaf - 1] = current;
    current++;
    currentLeaf -= 2;
    int32_t root = HUF_MAX_SYMBOL_COUNT + lastNonZero - 1;
    for (int32_t n = current; n <= root; n++) {
        nt->count[n] = 1 << 30;
    }
    while (current <= root) {
        int32_t child1, child2;
        if (currentLeaf >= 0 && nt->count[currentLeaf] < nt->count[currentNonLeaf]) {
            child1 = currentLeaf--;
        }
        else {
            child1 = currentNonLeaf++;
        }
        if (currentLeaf >= 0 && nt->count[currentLeaf] < nt->count[currentNonLeaf]) {
            child2 = currentLeaf--;
        }
        else {
            child2 = currentNonLeaf++;
        }
        nt->count[current] = nt->count[child1] + nt->count[child2];
        nt->parents[child1] = current;
        nt->parents[child2] = current;
        current++;
    }
    nt->numberOfBits[root] = 0;
    for (int32_t n = root - 1; n >= nonLeafStart; n--) {
        int16_t parent = nt->parents[n];
        nt->numberOfBits[n] = (uint8_t)(nt->numberOfBits[parent] + 1);
    }
    for (int32_t n = 0; n <= lastNonZero; n++) {
        int16_t parent = nt->parents[n];
        nt->numberOfBits[n] = (uint8_t)(nt->numberOfBits[parent] + 1);
    }
    return lastNonZero;
}

static int32_t huf_set_max_height(node_table* nt, int32_t lastNonZero, int32_t maxNumberOfBits, int32_t* rankLast) /* :294-390 */
{
    int32_t largestBits = nt->numberOfBits[lastNonZero];
    if (largestBits <= maxNumberOfBits) {
        return largestBits;
    }
    int32_t totalCost = 0;
    int32_t baseCost = 1 << (largestBits - maxNumberOfBits);
    int32_t n = lastNonZero;
    while (nt->numberOfBits[n] > maxNumberOfBits) {
        totalCost += baseCost - (1 << (largestBits - nt->numberOfBits[n]));
        nt->numberOfBits[n] = (uint8_t)maxNumberOfBits;
        n--;
    }
    while (nt->numberOfBits[n] == maxNumberOfBits) {
        n--;
    }
    totalCost = (int32_t)((uint32_t)totalCost >> (largestBits - maxNumberOfBits));

    const int32_t noSymbol = (int32_t)0xF0F0F0F0;
    for (int i = 0; i < HUF_MAX_TABLE_LOG + 2; i++) rankLast[i] = noSymbol;
    int32_t currentNbBits = maxNumberOfBits;
    for (int32_t pos = n; pos >= 0; pos--) {
        if (nt->numberOfBits[pos] >= currentNbBits) {
            continue;
        }
        currentNbBits = nt->numberOfBits[pos];
        rankLast[maxNumberOfBits - currentNbBits] = pos;
    }
    while (totalCost > 0) {
        int32_t numberOfBitsToDecrease = highest_bit((uint32_t)totalCost) + 1;
        for (; numberOfBitsToDecrease > 1; numberOfBitsToDecrease--) {
            int32_t highPosition = rankLast[numberOfBitsToDecrease];
            int32_t lowPosition = rankLast[numberOfBitsToDecrease - 1];
            if (highPosition == noSymbol) {
                continue;
            }
            if (lowPosition == noSymbol) {
                break;
            }
            int32_t highTotal = nt->count[highPosition];
            int32_t lowTotal = 2 * nt->count[lowPosition];
            if (highTotal <= lowTotal) {
                break;
            }
        }
        while ((numberOfBitsToDecrease <= HUF_MAX_TABLE_LOG) && (rankLast[numberOfBitsToDecrease] == noSymbol)) {
            numberOfBitsToDecrease++;
        }
        totalCost -= 1 << (numberOfBitsToDecrease - 1);
        if (rankLast[numberOfBitsToDecrease - 1] == noSymbol) {
            rankLast[numberOfBitsToDecrease - 1] = rankLast[numberOfBitsToDecrease];
        }
        nt->numberOfBits[rankLast[numberOfBitsToDecrease]]++;
        if (rankLast[numberOfBitsToDecrease] == 0) {
            rankLast[numberOfBitsToDecrease] = noSymbol;
        }
        else {
            rankLast[numberOfBitsToDecrease]--;
            if (nt->numberOfBits[rankLast[numberOfBitsToDecrease]] != maxNumberOfBits - numberOfBitsToDecrease) {
                rankLast[numberOfBitsToDecrease] = noSymbol;
            }
        }
    }
    while (totalCost < 0) {
        if (rankLast[1] == noSymbol) {
            while (nt->numberOfBits[n] == maxNumberOfBits) {
                n--;
            }
            nt->numberOfBits[n + 1]--;
            rankLast[1] = n + 1;
            totalCost++;
            continue;
        }
        nt->numberOfBits[rankLast[1] + 1]--;
        rankLast[1]++;
        totalCost++;
    }
    return maxNumberOfBits;
}

static void huf_table_initialize(huf_ctable* t, const int32_t* counts, int32_t maxSymbol, int32_t maxNumberOfBits, huf_context* ws) /* :60-103 */
{
    memset(ws->entriesPerRank, 0, sizeof(ws->entriesPerRank));
    memset(ws->valuesPerRank, 0, sizeof(ws->valuesPerRank));
    node_table* nt = &ws->nodeTable;
    memset(nt, 0, sizeof(*nt));
    int32_t lastNonZero = huf_build_tree(counts, maxSymbol, nt);
    maxNumberOfBits = huf_set_max_height(nt, lastNonZero, maxNumberOfBits, ws->rankLast);

    int32_t symbolCount = maxSymbol + 1;
    for (int32_t node = 0; node < symbolCount; node++) {
        int32_t symbol = nt->symbols[node];
        t->numberOfBits[symbol] = nt->numberOfBits[node];
    }
    for (int32_t n = 0; n <= lastNonZero; n++) {
        ws->entriesPerRank[nt->numberOfBits[n]]++;
    }
    int16_t startingValue = 0;
    for (int32_t rank = maxNumberOfBits; rank > 0; rank--) {
        ws->valuesPerRank[rank] = startingValue;
        startingValue = (int16_t)(startingValue + ws->entriesPerRank[rank]);
        startingValue = (int16_t)((uint32_t)(int32_t)startingValue >> 1); /* short >>>= 1 : int promotion, then narrowing */
    }
    for (int32_t n = 0; n <= maxSymbol; n++) {
        t->values[n] = ws->valuesPerRank[t->numberOfBits[n]]++;
    }
    t->maxSymbol = maxSymbol;
    t->maxNumberOfBits = maxNumberOfBits;
}

/* HuffmanCompressionTable.compressWeights :392-436 */
static int32_t huf_compress_weights(uint8_t* base, int64_t outputAddress, int32_t outputSize, const uint8_t* weights, int32_t weightsLength, huf_context* ws)
{
    if (weightsLength <= 1) {
        return 0;
    }
    histogram_count(weights, weightsLength, ws->wcounts, HUF_MAX_TABLE_LOG + 1);
    int32_t maxSymbol = find_max_symbol(ws->wcounts, HUF_MAX_TABLE_LOG);
    int32_t maxCount = find_largest_count(ws->wcounts, maxSymbol);
    if (maxCount == weightsLength) {
        return 1;
    }
    if (maxCount == 1) {
        return 0;
    }
    int32_t tableLog = fse_optimal_table_log(HUF_MAX_FSE_TABLE_LOG, weightsLength, maxSymbol);
    fse_normalize_counts(ws->wnorm, tableLog, ws->wcounts, weightsLength, maxSymbol);
    int64_t output = outputAddress;
    int64_t outputLimit = outputAddress + outputSize;
    int32_t headerSize = fse_write_normalized_counts(base, output, outputSize, ws->wnorm, maxSymbol, tableLog);
    output += headerSize;
    fse_initialize(&ws->wfse, ws->wnorm, maxSymbol, tableLog);
    int32_t compressedSize = fse_compress(base, output, (int32_t)(outputLimit - output), weights, weightsLength, &ws->wfse);
    if (compressedSize == 0) {
        return 0;
    }
    output += compressedSize;
    return (int32_t)(output - outputAddress);
}

/* HuffmanCompressionTable.write :206-268 */
static int32_t huf_table_write(const huf_ctable* t, uint8_t* base, int64_t outputAddress, int32_t outputSize, huf_context* ws)
{
    uint8_t* weights = ws->weights;
    int64_t output = outputAddress;
    int32_t maxNumberOfBits = t->maxNumberOfBits;
    int32_t maxSymbol = t->maxSymbol;
    for (int32_t symbol = 0; symbol < maxSymbol; symbol++) {
        int32_t bits = t->numberOfBits[symbol];
        weights[symbol] = bits == 0 ? 0 : (uint8_t)(maxNumberOfBits + 1 - bits);
    }
    int32_t size = huf_compress_weights(base, output + 1, outputSize - 1, weights, maxSymbol, ws);
    if (maxSymbol > 127 && size > 127) {
        fail(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED); /* Java: AssertionError */
    }
    if (size != 0 && size != 1 && size < maxSymbol / 2) {
        base[output] = (uint8_t)size;
        return size + 1;
    }
    int32_t entryCount = maxSymbol;
    size = (entryCount + 1) / 2;
    CHECK_ARGUMENT(size + 1 <= outputSize);
    base[output] = (uint8_t)(127 + entryCount);
    output++;
    if (maxSymbol >= HUF_MAX_SYMBOL) {
        fail(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED); /* Java: weights[255] ArrayIndexOutOfBounds */
    }
    weights[maxSymbol] = 0;
    for (int32_t i = 0; i < entryCount; i += 2) {
        base[output] = (uint8_t)((weights[i] << 4) + weights[i + 1]);
        output++;
    }
    return (int32_t)(output - outputAddress);
}

static int huf_table_is_valid(const huf_ctable* t, const int32_t* counts, int32_t maxSymbol) /* :273-286 */
{
    if (maxSymbol > t->maxSymbol) {
        return 0;
    }
    for (int32_t symbol = 0; symbol <= maxSymbol; ++symbol) {
        if (counts[symbol] != 0 && t->numberOfBits[symbol] == 0) {
            return 0;
        }
    }
    return 1;
}
static int32_t huf_estimate_compressed_size(const huf_ctable* t, const int32_t* counts, int32_t maxSymbol) /* :288-296 */
{
    int32_t numberOfBits = 0;
    int32_t lim = maxSymbol < t->maxSymbol ? maxSymbol : t->maxSymbol;
    for (int32_t symbol = 0; symbol <= lim; symbol++) {
        numberOfBits += t->numberOfBits[symbol] * counts[symbol];
    }
    return (int32_t)((uint32_t)numberOfBits >> 3);
}

/* HuffmanCompressor.compressSingleStream :88-134 */
static int32_t huf_compress_single_stream(uint8_t* base, int64_t outputAddress, int32_t outputSize, const uint8_t* in, int32_t inputSize, const huf_ctable* t)
{
    if (outputSize < SIZE_OF_LONG) {
        return 0;
    }
    bitout bs;
    bo_init(&bs, base, outputAddress, outputSize);
    int32_t n = inputSize & ~3;
#define ENC(sym) bo_add_bits_fast(&bs, t->values[(sym)], t->numberOfBits[(sym)])
    switch (inputSize & 3) {
        case 3: ENC(in[n + 2]); /* fallthrough */ /* 64 < 12*4+7 is false: no flush */
        case 2: ENC(in[n + 1]); /* fallthrough */
        case 1: ENC(in[n + 0]); bo_flush(&bs); /* fallthrough */
        default: break;
    }
    for (; n > 0; n -= 4) {
        ENC(in[n - 1]);
        ENC(in[n - 2]);
        ENC(in[n - 3]);
        ENC(in[n - 4]);
        bo_flush(&bs);
    }
#undef ENC
    return bo_close(&bs);
}

/* HuffmanCompressor.compress4streams :26-86 */
static int32_t huf_compress_4streams(uint8_t* base, int64_t outputAddress, int32_t outputSize, const uint8_t* in, int32_t inputSize, const huf_ctable* t)
{
    int64_t output = outputAddress;
    int64_t outputLimit = outputAddress + outputSize;
    int32_t segmentSize = (inputSize + 3) / 4;
    if (outputSize < 6 + 1 + 1 + 1 + 8) {
        return 0;
    }
    if (inputSize <= 6 + 1 + 1 + 1) {
        return 0;
    }
    output += 6;
    int32_t input = 0;
    int32_t compressedSize;
    for (int k = 0; k < 3; k++) {
        compressedSize = huf_compress_single_stream(base, output, (int32_t)(outputLimit - output), in + input, segmentSize, t);
        if (compressedSize == 0) {
            return 0;
        }
        st16(base + outputAddress + 2 * k, (uint32_t)compressedSize);
        output += compressedSize;
        input += segmentSize;
    }
    compressedSize = huf_compress_single_stream(base, output, (int32_t)(outputLimit - output), in + input, inputSize - input, t);
    if (compressedSize == 0) {
        return 0;
    }
    output += compressedSize;
    return (int32_t)(output - outputAddress);
}

/* ---- compression context ---- */
typedef struct {
    int32_t windowLog, windowSize, blockSize, chainLog, hashLog, searchLog, searchLength, targetLength;
} cparams;

typedef struct {
    uint8_t* literalsBuffer;
    int32_t literalsLength;
    int32_t* offsets;
    int32_t* literalLengths;
    int32_t* matchLengths;
    int32_t sequenceCount;
    uint8_t* literalLengthCodes;
    uint8_t* matchLengthCodes;
    uint8_t* offsetCodes;
    int32_t longLengthField; /* 0 none, 1 literal, 2 match */
    int32_t longLengthPosition;
} seq_store;

typedef struct {
    cparams p;
    /* RepeatedOffsets */
    int32_t offset0, offset1, tempOffset0, tempOffset1;
    /* BlockCompressionState */
    int32_t* hashTable;
    int32_t* chainTable;
    int32_t windowBaseOffset;
    seq_store ss;
    /* SequenceEncodingContext */
    fse_ctable llTable, ofTable, mlTable;
    int32_t counts[MAX_MATCH_LENGTH_SYMBOL + 1];
    int16_t normalizedCounts[MAX_MATCH_LENGTH_SYMBOL + 1];
    huf_context huf;
} cctx;

static const int32_t LEVEL3[4][7] = {
    {20, 16, 17, 1, 5, 1, 0}, /* default */
    {18, 16, 16, 1, 4, 1, 0}, /* <= 256 KB */
    {17, 15, 16, 2, 5, 1, 0}, /* <= 128 KB */
    {14, 14, 14, 2, 4, 1, 0}, /* <= 16 KB */
};

/* CompressionParameters.compute :256-299 for level 3 (strategy DFAST: cycleLog == chainLog) */
static cparams compute_parameters(int32_t estimatedInputSize)
{
    int table = 0;
    if (estimatedInputSize <= 16 * 1024) table = 3;
    else if (estimatedInputSize <= 128 * 1024) table = 2;
    else if (estimatedInputSize <= 256 * 1024) table = 1;
    int32_t windowLog = LEVEL3[table][0], chainLog = LEVEL3[table][1], hashLog = LEVEL3[table][2];
    int32_t searchLog = LEVEL3[table][3], searchLength = LEVEL3[table][4], targetLength = LEVEL3[table][5];
    /* estimatedInputSize < 1 << 30 always holds for int sizes below 2^30; Java compares against 1L << 30 */
    if ((int64_t)estimatedInputSize < (1LL << (MAX_WINDOW_LOG - 1))) {
        int32_t hashSizeMin = 1 << 6;
        int32_t inputSizeLog = (estimatedInputSize < hashSizeMin) ? 6 : highest_bit((uint32_t)(estimatedInputSize - 1)) + 1;
        if (windowLog > inputSizeLog) {
            windowLog = inputSizeLog;
        }
    }
    if (hashLog > windowLog + 1) {
        hashLog = windowLog + 1;
    }
    int32_t cycleLog = chainLog;
    if (cycleLog > windowLog) {
        chainLog -= (cycleLog - windowLog);
    }
    if (windowLog < MIN_WINDOW_LOG) {
        windowLog = MIN_WINDOW_LOG;
    }
    cparams p;
    p.windowLog = windowLog;
    p.windowSize = 1 << windowLog;
    p.blockSize = p.windowSize < MAX_BLOCK_SIZE ? p.windowSize : MAX_BLOCK_SIZE;
    p.chainLog = chainLog;
    p.hashLog = hashLog;
    p.searchLog = searchLog;
    p.searchLength = searchLength;
    p.targetLength = targetLength;
    return p;
}

/* ---- SequenceStore ---- */
static const uint8_t LITERAL_LENGTH_CODE[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20, 20, 21, 21, 21, 21,
                                                22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
static const uint8_t MATCH_LENGTH_CODE[128] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
                                               32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39,
                                               40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41,
                                               42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};

static void store_sequence(seq_store* s, const uint8_t* in, int64_t literalAddress, int32_t literalLength, int32_t offsetCode, int32_t matchLengthBase) /* :83-111 */
{
    memcpy(s->literalsBuffer + s->literalsLength, in + literalAddress, (size_t)literalLength); /* the 8-byte over-copy only touches bytes rewritten later */
    s->literalsLength += literalLength;
    if (literalLength > 65535) {
        s->longLengthField = 1;
        s->longLengthPosition = s->sequenceCount;
    }
    s->literalLengths[s->sequenceCount] = literalLength;
    s->offsets[s->sequenceCount] = offsetCode + 1;
    if (matchLengthBase > 65535) {
        s->longLengthField = 2;
        s->longLengthPosition = s->sequenceCount;
    }
    s->matchLengths[s->sequenceCount] = matchLengthBase;
    s->sequenceCount++;
}

static void generate_codes(seq_store* s) /* :121-135 */
{
    for (int32_t i = 0; i < s->sequenceCount; ++i) {
        int32_t ll = s->literalLengths[i];
        s->literalLengthCodes[i] = (uint8_t)(ll >= 64 ? highest_bit((uint32_t)ll) + 19 : LITERAL_LENGTH_CODE[ll]);
        s->offsetCodes[i] = (uint8_t)highest_bit((uint32_t)s->offsets[i]);
        int32_t ml = s->matchLengths[i];
        s->matchLengthCodes[i] = (uint8_t)(ml >= 128 ? highest_bit((uint32_t)ml) + 36 : MATCH_LENGTH_CODE[ml]);
    }
    if (s->longLengthField == 1) {
        s->literalLengthCodes[s->longLengthPosition] = MAX_LITERALS_LENGTH_SYMBOL;
    }
    if (s->longLengthField == 2) {
        s->matchLengthCodes[s->longLengthPosition] = MAX_MATCH_LENGTH_SYMBOL;
    }
}

/* ---- DoubleFastBlockCompressor ---- */
static inline int32_t hash4(uint32_t value, int32_t bits) { return (int32_t)((value * 0x9E3779B1u) >> (32 - bits)); }
static inline int32_t hash5(uint64_t value, int32_t bits) { return (int32_t)(((value << (64 - 40)) * 0xCF1BBCDCBBULL) >> (64 - bits)); }
static inline int32_t hash6(uint64_t value, int32_t bits) { return (int32_t)(((value << (64 - 48)) * 0xCF1BBCDCBF9BULL) >> (64 - bits)); }
static inline int32_t hash7(uint64_t value, int32_t bits) { return (int32_t)(((value << (64 - 56)) * 0xCF1BBCDCBFA563ULL) >> (64 - bits)); }
static inline int32_t hash8(uint64_t value, int32_t bits) { return (int32_t)((value * 0xCF1BBCDCB7A56463ULL) >> (64 - bits)); }
static inline int32_t hash_n(const uint8_t* in, int64_t address, int32_t bits, int32_t matchSearchLength) /* :213-222 */
{
    switch (matchSearchLength) {
        case 8: return hash8(ld64(in + address), bits);
        case 7: return hash7(ld64(in + address), bits);
        case 6: return hash6(ld64(in + address), bits);
        case 5: return hash5(ld64(in + address), bits);
        default: return hash4(ld32(in + address), bits);
    }
}
static int32_t dfast_count(const uint8_t* in, int64_t inputAddress, int64_t inputLimit, int64_t matchAddress) /* :183-211 */
{
    int64_t input = inputAddress, match = matchAddress;
    int32_t remaining = (int32_t)(inputLimit - inputAddress);
    int32_t count = 0;
    while (count < remaining - 7) {
        uint64_t diff = ld64(in + match) ^ ld64(in + input);
        if (diff != 0) {
            return count + (__builtin_ctzll(diff) >> 3);
        }
        count += 8;
        input += 8;
        match += 8;
    }
    while (count < remaining && in[match] == in[input]) {
        count++;
        input++;
        match++;
    }
    return count;
}

/* compressBlock :28-180.  Addresses are positions in `in` (baseAddress = 0 = frame start). */
static int32_t dfast_compress_block(cctx* c, const uint8_t* in, int64_t inputAddress, int32_t inputSize)
{
    const int32_t MIN_MATCH = 3, SEARCH_STRENGTH = 8, REP_MOVE = 2;
    int32_t matchSearchLength = c->p.searchLength > 4 ? c->p.searchLength : 4;
    const int64_t baseAddress = 0;
    const int64_t windowBaseAddress = baseAddress + c->windowBaseOffset;
    int32_t* longHashTable = c->hashTable;
    int32_t longHashBits = c->p.hashLog;
    int32_t* shortHashTable = c->chainTable;
    int32_t shortHashBits = c->p.chainLog;
    const int64_t inputEnd = inputAddress + inputSize;
    const int64_t inputLimit = inputEnd - SIZE_OF_LONG;
    int64_t input = inputAddress;
    int64_t anchor = inputAddress;
    int32_t offset1 = c->offset0;
    int32_t offset2 = c->offset1;
    int32_t savedOffset = 0;
    if (input - windowBaseAddress == 0) {
        input++;
    }
    int32_t maxRep = (int32_t)(input - windowBaseAddress);
    if (offset2 > maxRep) {
        savedOffset = offset2;
        offset2 = 0;
    }
    if (offset1 > maxRep) {
        savedOffset = offset1;
        offset1 = 0;
    }
    while (input < inputLimit) {
        int32_t shortHash = hash_n(in, input, shortHashBits, matchSearchLength);
        int64_t shortMatchAddress = baseAddress + shortHashTable[shortHash];
        int32_t longHash = hash8(ld64(in + input), longHashBits);
        int64_t longMatchAddress = baseAddress + longHashTable[longHash];
        int32_t current = (int32_t)(input - baseAddress);
        longHashTable[longHash] = current;
        shortHashTable[shortHash] = current;
        int32_t matchLength;
        int32_t offset;
        if (offset1 > 0 && ld32(in + input + 1 - offset1) == ld32(in + input + 1)) {
            matchLength = dfast_count(in, input + 1 + 4, inputEnd, input + 1 + 4 - offset1) + 4;
            input++;
            store_sequence(&c->ss, in, anchor, (int32_t)(input - anchor), 0, matchLength - MIN_MATCH);
        }
        else {
            if (longMatchAddress > windowBaseAddress && ld64(in + longMatchAddress) == ld64(in + input)) {
                matchLength = dfast_count(in, input + 8, inputEnd, longMatchAddress + 8) + 8;
                offset = (int32_t)(input - longMatchAddress);
                while (input > anchor && longMatchAddress > windowBaseAddress && in[input - 1] == in[longMatchAddress - 1]) {
                    input--;
                    longMatchAddress--;
                    matchLength++;
                }
            }
            else {
                if (shortMatchAddress > windowBaseAddress && ld32(in + shortMatchAddress) == ld32(in + input)) {
                    int32_t nextOffsetHash = hash8(ld64(in + input + 1), longHashBits);
                    int64_t nextOffsetMatchAddress = baseAddress + longHashTable[nextOffsetHash];
                    longHashTable[nextOffsetHash] = current + 1;
                    if (nextOffsetMatchAddress > windowBaseAddress && ld64(in + nextOffsetMatchAddress) == ld64(in + input + 1)) {
                        matchLength = dfast_count(in, input + 1 + 8, inputEnd, nextOffsetMatchAddress + 8) + 8;
                        input++;
                        offset = (int32_t)(input - nextOffsetMatchAddress);
                        while (input > anchor && nextOffsetMatchAddress > windowBaseAddress && in[input - 1] == in[nextOffsetMatchAddress - 1]) {
                            input--;
                            nextOffsetMatchAddress--;
                            matchLength++;
                        }
                    }
                    else {
                        matchLength = dfast_count(in, input + 4, inputEnd, shortMatchAddress + 4) + 4;
                        offset = (int32_t)(input - shortMatchAddress);
                        while (input > anchor && shortMatchAddress > windowBaseAddress && in[input - 1] == in[shortMatchAddress - 1]) {
                            input--;
                            shortMatchAddress--;
                            matchLength++;
                        }
                    }
                }
                else {
                    input += ((input - anchor) >> SEARCH_STRENGTH) + 1;
                    continue;
                }
            }
            offset2 = offset1;
            offset1 = offset;
            store_sequence(&c->ss, in, anchor, (int32_t)(input - anchor), offset + REP_MOVE, matchLength - MIN_MATCH);
        }
        input += matchLength;
        anchor = input;
        if (input <= inputLimit) {
            longHashTable[hash8(ld64(in + baseAddress + current + 2), longHashBits)] = current + 2;
            shortHashTable[hash_n(in, baseAddress + current + 2, shortHashBits, matchSearchLength)] = current + 2;
            longHashTable[hash8(ld64(in + input - 2), longHashBits)] = (int32_t)(input - 2 - baseAddress);
            shortHashTable[hash_n(in, input - 2, shortHashBits, matchSearchLength)] = (int32_t)(input - 2 - baseAddress);
            while (input <= inputLimit && offset2 > 0 && ld32(in + input) == ld32(in + input - offset2)) {
                int32_t repetitionLength = dfast_count(in, input + 4, inputEnd, input + 4 - offset2) + 4;
                int32_t temp = offset2;
                offset2 = offset1;
                offset1 = temp;
                shortHashTable[hash_n(in, input, shortHashBits, matchSearchLength)] = (int32_t)(input - baseAddress);
                longHashTable[hash8(ld64(in + input), longHashBits)] = (int32_t)(input - baseAddress);
                store_sequence(&c->ss, in, anchor, 0, 0, repetitionLength - MIN_MATCH);
                input += repetitionLength;
                anchor = input;
            }
        }
    }
    c->tempOffset0 = offset1 != 0 ? offset1 : savedOffset;
    c->tempOffset1 = offset2 != 0 ? offset2 : savedOffset;
    return (int32_t)(inputEnd - anchor);
}

/* ---- ZstdFrameCompressor: literals ---- */
static int32_t raw_literals(uint8_t* base, int64_t outputAddress, int32_t outputSize, const uint8_t* in, int32_t inputSize) /* :407-431 */
{
    int32_t headerSize = 1;
    if (inputSize >= 32) headerSize++;
    if (inputSize >= 4096) headerSize++;
    CHECK_ARGUMENT(inputSize + headerSize <= outputSize);
    switch (headerSize) {
        case 1: base[outputAddress] = (uint8_t)(RAW_LITERALS_BLOCK | (inputSize << 3)); break;
        case 2: st16(base + outputAddress, (uint32_t)(RAW_LITERALS_BLOCK | (1 << 2) | (inputSize << 4))); break;
        default: st24(base + outputAddress, (uint32_t)(RAW_LITERALS_BLOCK | (3 << 2) | (inputSize << 4))); break;
    }
    CHECK_ARGUMENT(inputSize + 1 <= outputSize);
    memcpy(base + outputAddress + headerSize, in, (size_t)inputSize);
    return headerSize + inputSize;
}

static int32_t rle_literals(uint8_t* base, int64_t outputAddress, const uint8_t* in, int32_t inputSize) /* :380-398 */
{
    int32_t headerSize = 1 + (inputSize > 31 ? 1 : 0) + (inputSize > 4095 ? 1 : 0);
    switch (headerSize) {
        case 1: base[outputAddress] = (uint8_t)(RLE_LITERALS_BLOCK | (inputSize << 3)); break;
        case 2: st16(base + outputAddress, (uint32_t)(RLE_LITERALS_BLOCK | (1 << 2) | (inputSize << 4))); break;
        default: st32(base + outputAddress, (uint32_t)(RLE_LITERALS_BLOCK | 3 << 2 | inputSize << 4)); break;
    }
    base[outputAddress + headerSize] = in[0];
    return headerSize + 1;
}

static int32_t calculate_minimum_gain(int32_t inputSize) { return (int32_t)((uint32_t)inputSize >> 6) + 2; } /* :400-405, strategy DFAST */

static int32_t encode_literals(cctx* c, uint8_t* base, int64_t outputAddress, int32_t outputSize, const uint8_t* literals, int32_t literalsSize) /* :262-378 */
{
    huf_context* h = &c->huf;
    if (literalsSize <= MINIMUM_LITERALS_SIZE) { /* bypassCompression is false: strategy is DFAST */
        return raw_literals(base, outputAddress, outputSize, literals, literalsSize);
    }
    int32_t headerSize = 3 + (literalsSize >= 1024 ? 1 : 0) + (literalsSize >= 16384 ? 1 : 0);
    CHECK_ARGUMENT(headerSize + 1 <= outputSize);
    int32_t counts[HUF_MAX_SYMBOL_COUNT];
    histogram_count(literals, literalsSize, counts, HUF_MAX_SYMBOL_COUNT);
    int32_t maxSymbol = find_max_symbol(counts, HUF_MAX_SYMBOL);
    int32_t largestCount = find_largest_count(counts, maxSymbol);
    if (largestCount == literalsSize) {
        return rle_literals(base, outputAddress, literals, literalsSize);
    }
    else if (largestCount <= (int32_t)((uint32_t)literalsSize >> 7) + 4) {
        return raw_literals(base, outputAddress, outputSize, literals, literalsSize);
    }
    huf_ctable* previousTable = &h->tables[h->previousTable];
    huf_ctable* table;
    int32_t serializedTableSize;
    int reuseTable;
    int canReuse = huf_table_is_valid(previousTable, counts, maxSymbol);
    int preferReuse = literalsSize <= 1024; /* DFAST.ordinal() < LAZY.ordinal() */
    if (preferReuse && canReuse) {
        table = previousTable;
        reuseTable = 1;
        serializedTableSize = 0;
    }
    else {
        /* borrowTemporaryTable */
        h->previousCandidate = h->temporaryTable;
        h->temporaryCandidate = h->previousTable;
        huf_ctable* newTable = &h->tables[h->temporaryTable];
        huf_table_initialize(newTable, counts, maxSymbol, huf_optimal_number_of_bits(MAX_HUFFMAN_TABLE_LOG, literalsSize, maxSymbol), h);
        serializedTableSize = huf_table_write(newTable, base, outputAddress + headerSize, outputSize - headerSize, h);
        if (canReuse && huf_estimate_compressed_size(previousTable, counts, maxSymbol) <= serializedTableSize + huf_estimate_compressed_size(newTable, counts, maxSymbol)) {
            table = previousTable;
            reuseTable = 1;
            serializedTableSize = 0;
            h->previousCandidate = h->previousTable; /* discardTemporaryTable */
            h->temporaryCandidate = h->temporaryTable;
        }
        else {
            table = newTable;
            reuseTable = 0;
        }
    }
    int32_t compressedSize;
    int singleStream = literalsSize < 256;
    if (singleStream) {
        compressedSize = huf_compress_single_stream(base, outputAddress + headerSize + serializedTableSize, outputSize - headerSize - serializedTableSize, literals, literalsSize, table);
    }
    else {
        compressedSize = huf_compress_4streams(base, outputAddress + headerSize + serializedTableSize, outputSize - headerSize - serializedTableSize, literals, literalsSize, table);
    }
    int32_t totalSize = serializedTableSize + compressedSize;
    int32_t minimumGain = calculate_minimum_gain(literalsSize);
    if (compressedSize == 0 || totalSize >= literalsSize - minimumGain) {
        h->previousCandidate = h->previousTable;
        h->temporaryCandidate = h->temporaryTable;
        return raw_literals(base, outputAddress, outputSize, literals, literalsSize);
    }
    int32_t encodingType = reuseTable ? TREELESS_LITERALS_BLOCK : COMPRESSED_LITERALS_BLOCK;
    switch (headerSize) {
        case 3: {
            uint32_t header = (uint32_t)(encodingType | ((singleStream ? 0 : 1) << 2) | (literalsSize << 4) | (totalSize << 14));
            st24(base + outputAddress, header);
            break;
        }
        case 4: {
            uint32_t header = (uint32_t)encodingType | (2u << 2) | ((uint32_t)literalsSize << 4) | ((uint32_t)totalSize << 18);  /* (Java's int arithmetic wraps: unsigned here) */
            st32(base + outputAddress, header);
            break;
        }
        default: {
            uint32_t header = (uint32_t)encodingType | (3u << 2) | ((uint32_t)literalsSize << 4) | ((uint32_t)totalSize << 22);
            st32(base + outputAddress, header);
            base[outputAddress + 4] = (uint8_t)((uint32_t)totalSize >> 10);
            break;
        }
    }
    return headerSize + totalSize;
}

/* ---- SequenceEncoder ---- */
static const int16_t DEFAULT_LL_NORM[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t DEFAULT_ML_NORM[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t DEFAULT_OF_NORM[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static fse_ctable g_default_ll, g_default_ml, g_default_of;
static int g_defaults_ready;

static int32_t select_encoding_type(int32_t largestCount, int32_t sequenceCount, int32_t defaultNormalizedCountsLog, int isDefaultTableAllowed) /* :299-341, strategy DFAST (ordinal 1) */
{
    if (largestCount == sequenceCount) {
        if (isDefaultTableAllowed && sequenceCount <= 2) {
            return SEQUENCE_ENCODING_BASIC;
        }
        return SEQUENCE_ENCODING_RLE;
    }
    if (isDefaultTableAllowed) {
        int32_t factor = 10 - 1;
        int32_t baseLog = 3;
        int64_t minNumberOfSequences = ((1LL << defaultNormalizedCountsLog) * factor) >> baseLog;
        if ((sequenceCount < minNumberOfSequences) || (largestCount < (sequenceCount >> (defaultNormalizedCountsLog - 1)))) {
            return SEQUENCE_ENCODING_BASIC;
        }
    }
    return SEQUENCE_ENCODING_COMPRESSED;
}

static int32_t build_compression_table(cctx* c, fse_ctable* table, uint8_t* base, int64_t output, int64_t outputLimit, int32_t sequenceCount, int32_t maxTableLog,
                                       const uint8_t* codes, int32_t* counts, int32_t maxSymbol, int16_t* normalizedCounts) /* :211-226 */
{
    (void)c;
    int32_t tableLog = fse_optimal_table_log(maxTableLog, sequenceCount, maxSymbol);
    if (counts[codes[sequenceCount - 1]] > 1) {
        counts[codes[sequenceCount - 1]]--;
        sequenceCount--;
    }
    fse_normalize_counts(normalizedCounts, tableLog, counts, sequenceCount, maxSymbol);
    fse_initialize(table, normalizedCounts, maxSymbol, tableLog);
    return fse_write_normalized_counts(base, output, (int32_t)(outputLimit - output), normalizedCounts, maxSymbol, tableLog);
}

static int32_t encode_sequences(uint8_t* base, int64_t output, int64_t outputLimit, const fse_ctable* matchLengthTable, const fse_ctable* offsetsTable,
                                const fse_ctable* literalLengthTable, const seq_store* s) /* :228-297 */
{
    const uint8_t* matchLengthCodes = s->matchLengthCodes;
    const uint8_t* offsetCodes = s->offsetCodes;
    const uint8_t* literalLengthCodes = s->literalLengthCodes;
    bitout bs;
    bo_init(&bs, base, output, (int32_t)(outputLimit - output));
    int32_t sequenceCount = s->sequenceCount;
    int32_t matchLengthState = fse_begin(matchLengthTable, matchLengthCodes[sequenceCount - 1]);
    int32_t offsetState = fse_begin(offsetsTable, offsetCodes[sequenceCount - 1]);
    int32_t literalLengthState = fse_begin(literalLengthTable, literalLengthCodes[sequenceCount - 1]);
    bo_add_bits(&bs, s->literalLengths[sequenceCount - 1], LITERALS_LENGTH_BITS[literalLengthCodes[sequenceCount - 1]]);
    bo_add_bits(&bs, s->matchLengths[sequenceCount - 1], MATCH_LENGTH_BITS[matchLengthCodes[sequenceCount - 1]]);
    bo_add_bits(&bs, s->offsets[sequenceCount - 1], offsetCodes[sequenceCount - 1]);
    bo_flush(&bs);
    if (sequenceCount >= 2) {
        for (int32_t n = sequenceCount - 2; n >= 0; n--) {
            int32_t literalLengthCode = literalLengthCodes[n];
            int32_t offsetCode = offsetCodes[n];
            int32_t matchLengthCode = matchLengthCodes[n];
            int32_t literalLengthBits = LITERALS_LENGTH_BITS[literalLengthCode];
            int32_t offsetBits = offsetCode;
            int32_t matchLengthBits = MATCH_LENGTH_BITS[matchLengthCode];
            offsetState = fse_encode(offsetsTable, &bs, offsetState, offsetCode);
            matchLengthState = fse_encode(matchLengthTable, &bs, matchLengthState, matchLengthCode);
            literalLengthState = fse_encode(literalLengthTable, &bs, literalLengthState, literalLengthCode);
            if (offsetBits + matchLengthBits + literalLengthBits >= 64 - 7 - (LITERAL_LENGTH_TABLE_LOG + MATCH_LENGTH_TABLE_LOG + OFFSET_TABLE_LOG)) {
                bo_flush(&bs);
            }
            bo_add_bits(&bs, s->literalLengths[n], literalLengthBits);
            if (literalLengthBits + matchLengthBits > 24) {
                bo_flush(&bs);
            }
            bo_add_bits(&bs, s->matchLengths[n], matchLengthBits);
            if (offsetBits + matchLengthBits + literalLengthBits > 56) {
                bo_flush(&bs);
            }
            bo_add_bits(&bs, s->offsets[n], offsetBits);
            bo_flush(&bs);
        }
    }
    fse_finish(matchLengthTable, &bs, matchLengthState);
    fse_finish(offsetsTable, &bs, offsetState);
    fse_finish(literalLengthTable, &bs, literalLengthState);
    int32_t streamSize = bo_close(&bs);
    CHECK_ARGUMENT(streamSize > 0);
    return streamSize;
}

static int32_t compress_sequences(cctx* c, uint8_t* base, int64_t outputAddress, int32_t outputSize) /* :66-209 */
{
    seq_store* s = &c->ss;
    int64_t output = outputAddress;
    int64_t outputLimit = outputAddress + outputSize;
    CHECK_ARGUMENT(outputLimit - output > 3 + 1);
    int32_t sequenceCount = s->sequenceCount;
    if (sequenceCount < 0x7F) {
        base[output] = (uint8_t)sequenceCount;
        output++;
    }
    else if (sequenceCount < LONG_NUMBER_OF_SEQUENCES) {
        base[output] = (uint8_t)((uint32_t)sequenceCount >> 8 | 0x80);
        base[output + 1] = (uint8_t)sequenceCount;
        output += 2;
    }
    else {
        base[output] = 0xFF;
        output++;
        st16(base + output, (uint32_t)(sequenceCount - LONG_NUMBER_OF_SEQUENCES));
        output += 2;
    }
    if (sequenceCount == 0) {
        return (int32_t)(output - outputAddress);
    }
    int64_t headerAddress = output++;
    int32_t maxSymbol, largestCount;
    int32_t* counts = c->counts;
    const int32_t COUNTS_LEN = MAX_MATCH_LENGTH_SYMBOL + 1;

    /* literal lengths */
    histogram_count(s->literalLengthCodes, sequenceCount, counts, COUNTS_LEN);
    maxSymbol = find_max_symbol(counts, MAX_LITERALS_LENGTH_SYMBOL);
    largestCount = find_largest_count(counts, maxSymbol);
    int32_t literalsLengthEncodingType = select_encoding_type(largestCount, sequenceCount, 6, 1);
    const fse_ctable* literalLengthTable;
    switch (literalsLengthEncodingType) {
        case SEQUENCE_ENCODING_RLE:
            base[output] = s->literalLengthCodes[0];
            output++;
            fse_init_rle(&c->llTable, maxSymbol);
            literalLengthTable = &c->llTable;
            break;
        case SEQUENCE_ENCODING_BASIC:
            literalLengthTable = &g_default_ll;
            break;
        default:
            output += build_compression_table(c, &c->llTable, base, output, outputLimit, sequenceCount, LITERAL_LENGTH_TABLE_LOG, s->literalLengthCodes, counts, maxSymbol, c->normalizedCounts);
            literalLengthTable = &c->llTable;
            break;
    }

    /* offsets */
    histogram_count(s->offsetCodes, sequenceCount, counts, COUNTS_LEN);
    maxSymbol = find_max_symbol(counts, MAX_OFFSET_CODE_SYMBOL);
    largestCount = find_largest_count(counts, maxSymbol);
    int defaultAllowed = maxSymbol < DEFAULT_MAX_OFFSET_CODE_SYMBOL;
    int32_t offsetEncodingType = select_encoding_type(largestCount, sequenceCount, 5, defaultAllowed);
    const fse_ctable* offsetCodeTable;
    switch (offsetEncodingType) {
        case SEQUENCE_ENCODING_RLE:
            base[output] = s->offsetCodes[0];
            output++;
            fse_init_rle(&c->ofTable, maxSymbol);
            offsetCodeTable = &c->ofTable;
            break;
        case SEQUENCE_ENCODING_BASIC:
            offsetCodeTable = &g_default_of;
            break;
        default: /* the Java code passes output + outputSize as the limit here (:155) */
            output += build_compression_table(c, &c->ofTable, base, output, output + outputSize, sequenceCount, OFFSET_TABLE_LOG, s->offsetCodes, counts, maxSymbol, c->normalizedCounts);
            offsetCodeTable = &c->ofTable;
            break;
    }

    /* match lengths */
    histogram_count(s->matchLengthCodes, sequenceCount, counts, COUNTS_LEN);
    maxSymbol = find_max_symbol(counts, MAX_MATCH_LENGTH_SYMBOL);
    largestCount = find_largest_count(counts, maxSymbol);
    int32_t matchLengthEncodingType = select_encoding_type(largestCount, sequenceCount, 6, 1);
    const fse_ctable* matchLengthTable;
    switch (matchLengthEncodingType) {
        case SEQUENCE_ENCODING_RLE:
            base[output] = s->matchLengthCodes[0];
            output++;
            fse_init_rle(&c->mlTable, maxSymbol);
            matchLengthTable = &c->mlTable;
            break;
        case SEQUENCE_ENCODING_BASIC:
            matchLengthTable = &g_default_ml;
            break;
        default:
            output += build_compression_table(c, &c->mlTable, base, output, outputLimit, sequenceCount, MATCH_LENGTH_TABLE_LOG, s->matchLengthCodes, counts, maxSymbol, c->normalizedCounts);
            matchLengthTable = &c->mlTable;
            break;
    }

    base[headerAddress] = (uint8_t)((literalsLengthEncodingType << 6) | (offsetEncodingType << 4) | (matchLengthEncodingType << 2));
    output += encode_sequences(base, output, outputLimit, matchLengthTable, offsetCodeTable, literalLengthTable, s);
    return (int32_t)(output - outputAddress);
}

/* ---- ZstdFrameCompressor: blocks and frame ---- */
static int32_t compress_block(cctx* c, const uint8_t* in, int64_t inputAddress, int32_t inputSize, uint8_t* base, int64_t outputAddress, int32_t outputSize) /* :206-260 */
{
    if (inputSize < MIN_BLOCK_SIZE + SIZE_OF_BLOCK_HEADER + 1) {
        return 0;
    }
    /* enforceMaxDistance (BlockCompressionState.java:61-69), baseAddress = 0 */
    int32_t distance = (int32_t)(inputAddress + inputSize);
    int32_t newOffset = distance - c->p.windowSize;
    if (c->windowBaseOffset < newOffset) {
        c->windowBaseOffset = newOffset;
    }
    c->ss.literalsLength = 0;
    c->ss.sequenceCount = 0;
    c->ss.longLengthField = 0;

    int32_t lastLiteralsSize = dfast_compress_block(c, in, inputAddress, inputSize);
    int64_t lastLiteralsAddress = inputAddress + inputSize - lastLiteralsSize;
    memcpy(c->ss.literalsBuffer + c->ss.literalsLength, in + lastLiteralsAddress, (size_t)lastLiteralsSize);
    c->ss.literalsLength += lastLiteralsSize;
    generate_codes(&c->ss);

    int64_t outputLimit = outputAddress + outputSize;
    int64_t output = outputAddress;
    int32_t compressedLiteralsSize = encode_literals(c, base, output, (int32_t)(outputLimit - output), c->ss.literalsBuffer, c->ss.literalsLength);
    output += compressedLiteralsSize;
    int32_t compressedSequencesSize = compress_sequences(c, base, output, (int32_t)(outputLimit - output));
    int32_t compressedSize = compressedLiteralsSize + compressedSequencesSize;
    if (compressedSize == 0) {
        return compressedSize;
    }
    int32_t maxCompressedSize = inputSize - calculate_minimum_gain(inputSize);
    if (compressedSize > maxCompressedSize) {
        return 0;
    }
    /* context.commit() */
    c->offset0 = c->tempOffset0;
    c->offset1 = c->tempOffset1;
    c->huf.temporaryTable = c->huf.temporaryCandidate;
    c->huf.previousTable = c->huf.previousCandidate;
    return compressedSize;
}

static void build_defaults(void)
{
    if (g_defaults_ready) return;
    fse_initialize(&g_default_ll, DEFAULT_LL_NORM, MAX_LITERALS_LENGTH_SYMBOL, 6);
    fse_initialize(&g_default_ml, DEFAULT_ML_NORM, MAX_MATCH_LENGTH_SYMBOL, 6);
    fse_initialize(&g_default_of, DEFAULT_OF_NORM, DEFAULT_MAX_OFFSET_CODE_SYMBOL, 5);
    g_defaults_ready = 1;
}

/* writeFrameHeader :64-121 ; returns the header size, or -1 / -2 for the two IllegalArgumentExceptions (:88-95) */
static int32_t write_frame_header(uint8_t* out, int32_t inputSize, int32_t windowSize)
{
    int32_t output = 0;
    int32_t contentSizeDescriptor = 0;
    if (inputSize != -1) {
        contentSizeDescriptor = (inputSize >= 256 ? 1 : 0) + (inputSize >= 65536 + 256 ? 1 : 0);
    }
    int32_t frameHeaderDescriptor = (contentSizeDescriptor << 6) | 0x04;
    int singleSegment = inputSize != -1 && windowSize >= inputSize;
    if (singleSegment) {
        frameHeaderDescriptor |= 0x20;
    }
    out[output++] = (uint8_t)frameHeaderDescriptor;
    if (!singleSegment) {
        int32_t base = (int32_t)(0x80000000u >> __builtin_clz((uint32_t)windowSize));
        int32_t exponent = 32 - __builtin_clz((uint32_t)base) - 1;
        if (exponent < MIN_WINDOW_LOG) {
            return -1; /* "Minimum window size is 1024" */
        }
        int32_t remainder = windowSize - base;
        if (remainder % (base / 8) != 0) {
            return -2; /* "Window size of magnitude 2^e must be multiple of base/8" */
        }
        int32_t mantissa = remainder / (base / 8);
        int32_t encoded = ((exponent - MIN_WINDOW_LOG) << 3) | mantissa;
        out[output++] = (uint8_t)encoded;
    }
    switch (contentSizeDescriptor) {
        case 0:
            if (singleSegment) {
                out[output++] = (uint8_t)inputSize;
            }
            break;
        case 1:
            st16(out + output, (uint32_t)(inputSize - 256));
            output += 2;
            break;
        default:
            st32(out + output, (uint32_t)inputSize);
            output += 4;
            break;
    }
    return output;
}

/* test hook for the reference's frame-header KATs (T/zstd/TestCompressor.java:52-98) */
int32_t orc_zstd_write_frame_header(uint8_t* out14, int32_t inputSize, int32_t windowSize) { return write_frame_header(out14, inputSize, windowSize); }

static int64_t zstd_compress(cctx* c, const uint8_t* in, int32_t inputSize, uint8_t* out, int64_t outCap)
{
    build_defaults();
    const int64_t outputLimit = outCap;
    int64_t output = 0;
    c->p = compute_parameters(inputSize);

    /* writeMagic :55-61 */
    CHECK_ARGUMENT(outputLimit - output >= 4);
    st32(out + output, MAGIC_NUMBER);
    output += 4;

    CHECK_ARGUMENT(outputLimit - output >= MAX_FRAME_HEADER_SIZE);
    output += write_frame_header(out + output, inputSize, c->p.windowSize);

    /* compressFrame :152-179 with a fresh CompressionContext (:162) */
    {
        int32_t blockSize = c->p.blockSize;
        int32_t outputSize = (int32_t)(outputLimit - output);
        int32_t remaining = inputSize;
        int64_t input = 0;
        c->offset0 = 1;
        c->offset1 = 4;
        c->tempOffset0 = c->tempOffset1 = 0;
        c->windowBaseOffset = 0;
        memset(c->hashTable, 0, sizeof(int32_t) << c->p.hashLog);
        memset(c->chainTable, 0, sizeof(int32_t) << c->p.chainLog);
        memset(&c->huf.tables, 0, sizeof(c->huf.tables));
        c->huf.previousTable = 0;
        c->huf.temporaryTable = 1;
        c->huf.previousCandidate = 0;
        c->huf.temporaryCandidate = 1;
        do {
            CHECK_ARGUMENT(outputSize >= SIZE_OF_BLOCK_HEADER + MIN_BLOCK_SIZE);
            int lastBlock = blockSize >= remaining;
            blockSize = blockSize < remaining ? blockSize : remaining;
            /* writeCompressedBlock :181-204 */
            int32_t compressedSize = 0;
            if (blockSize > 0) {
                compressedSize = compress_block(c, in, input, blockSize, out, output + SIZE_OF_BLOCK_HEADER, outputSize - SIZE_OF_BLOCK_HEADER);
            }
            if (compressedSize == 0) {
                CHECK_ARGUMENT(blockSize + SIZE_OF_BLOCK_HEADER <= outputSize);
                int32_t blockHeader = (lastBlock ? 1 : 0) | (RAW_BLOCK << 1) | (blockSize << 3);
                st24(out + output, (uint32_t)blockHeader);
                if (blockSize > 0) {
                    memcpy(out + output + SIZE_OF_BLOCK_HEADER, in + input, (size_t)blockSize);
                }
                compressedSize = SIZE_OF_BLOCK_HEADER + blockSize;
            }
            else {
                int32_t blockHeader = (lastBlock ? 1 : 0) | (COMPRESSED_BLOCK << 1) | (compressedSize << 3);
                st24(out + output, (uint32_t)blockHeader);
                compressedSize += SIZE_OF_BLOCK_HEADER;
            }
            input += blockSize;
            remaining -= blockSize;
            output += compressedSize;
            outputSize -= compressedSize;
        }
        while (remaining > 0);
    }

    /* writeChecksum :123-134 */
    CHECK_ARGUMENT(outputLimit - output >= 4);
    st32(out + output, (uint32_t)orc_xxh64(in, inputSize, 0));
    output += 4;
    return output;
}

/*
 * ZstdOutputStream (M/zstd/ZstdOutputStream.java:30-221) for ONE write(buffer, 0, n) followed by close() -- the way the reference's
 * own stream harness drives it (T/HadoopCodecCompressor.java:57-72) -- restated without the buffer copies: the stream's buffer is a
 * sliding view of the input (`origin` = the input position of buffer index 0), everything the encoder sees is relative to it.
 *   :48-58    parameters for an UNKNOWN size (CompressionParameters.compute(3, -1) = the default row as it stands: window 1 MiB, hash 2^17,
 *             chain 2^16), buffer limit 4 x window = 4 MiB, a fresh CompressionContext
 *   :107-120  growBufferIfNecessary: the buffer becomes min(2 n, 4 MiB) (at least one block) once per write()
 *   :122-131  compressIfNecessary: only a FULL buffer of the maximum size is flushed -- so an input below 4 MiB is written by close()
 *             alone, as one chunk whose size the frame header announces
 *   :154-221  writeChunk: a flush writes whole blocks and keeps window + one block unprocessed (23 blocks the first time, 15 from then
 *             on), then slides: table entries move down by the slide (:35-49 of BlockCompressionState, clamped at 0) and the buffer's
 *             tail moves to its front.  The window base (BlockCompressionState.windowBaseOffset) does NOT move with them -- enforceMaxDistance
 *             (:61-69) only ever raises it -- so after a slide it lies AHEAD of the next blocks, which therefore find no match at all
 *             (every candidate fails `matchAddress > windowBaseAddress`, both repeat offsets are put aside) until the position has caught
 *             up with it: 7 blocks of literals after every slide.  Kept as it is: the bytes are the reference's.
 * n >= 2^30 is refused (the Java code computes (0 + n) * 2 in int: the buffer would be 128 KiB and write() would never return).
 */
static int64_t zstd_stream_compress(cctx* c, const uint8_t* src, int32_t n, uint8_t* out, int64_t outCap)
{
    build_defaults();
    /* :48-58 */
    c->p.windowLog = LEVEL3[0][0];
    c->p.windowSize = 1 << c->p.windowLog;
    c->p.blockSize = c->p.windowSize < MAX_BLOCK_SIZE ? c->p.windowSize : MAX_BLOCK_SIZE;
    c->p.chainLog = LEVEL3[0][1];
    c->p.hashLog = LEVEL3[0][2];
    c->p.searchLog = LEVEL3[0][3];
    c->p.searchLength = LEVEL3[0][4];
    c->p.targetLength = LEVEL3[0][5];
    const int32_t windowSize = c->p.windowSize, blockSizeMax = c->p.blockSize;
    const int32_t maxBufferSize = windowSize * 4;
    /* the per-block output buffer of the stream (:55-58): what writeCompressedBlock is told it may use */
    const int32_t compressedLength = (blockSizeMax + SIZE_OF_BLOCK_HEADER) + ((blockSizeMax + SIZE_OF_BLOCK_HEADER) >> 8) + SIZE_OF_LONG;
    static __thread uint8_t* compressed;
    if (!compressed) {
        compressed = (uint8_t*)malloc((size_t)compressedLength + 64);
    }
    c->offset0 = 1;
    c->offset1 = 4;
    c->tempOffset0 = c->tempOffset1 = 0;
    c->windowBaseOffset = 0;
    memset(c->hashTable, 0, sizeof(int32_t) << c->p.hashLog);
    memset(c->chainTable, 0, sizeof(int32_t) << c->p.chainLog);
    memset(&c->huf.tables, 0, sizeof(c->huf.tables));
    c->huf.previousTable = 0;
    c->huf.temporaryTable = 1;
    c->huf.previousCandidate = 0;
    c->huf.temporaryCandidate = 1;

    int64_t output = 0;
    int32_t bufferLength = 0;            /* uncompressed.length */
    int64_t origin = 0;                  /* input position of buffer index 0 */
    int32_t offset = 0, position = 0;    /* uncompressedOffset, uncompressedPosition */
    int firstChunk = 1;
    int32_t length = n;
    /* growBufferIfNecessary(n) :107-120 */
    if (!(position + length <= bufferLength || bufferLength >= maxBufferSize)) {
        int64_t newSize = ((int64_t)bufferLength + length) * 2;
        newSize = newSize < maxBufferSize ? newSize : maxBufferSize;
        newSize = newSize > blockSizeMax ? newSize : blockSizeMax;
        bufferLength = (int32_t)newSize;
    }
    for (int closing = 0; closing <= 1; closing++) {
        for (;;) {
            int flush;
            if (!closing) {
                if (length <= 0) {
                    break;
                }
                /* write :93-104 */
                int32_t writeSize = length < bufferLength - position ? length : bufferLength - position;
                position += writeSize;
                length -= writeSize;
                /* compressIfNecessary :122-131 */
                flush = bufferLength >= maxBufferSize && position == bufferLength && bufferLength - windowSize > blockSizeMax;
                if (!flush) {
                    continue;
                }
            }
            /* writeChunk(lastChunk = closing) :154-221 */
            int32_t chunkSize;
            if (closing) {
                chunkSize = position - offset;
            }
            else {
                chunkSize = position - offset - windowSize - blockSizeMax;
                chunkSize = (chunkSize / blockSizeMax) * blockSizeMax; /* (> one block: the buffer is 4 windows) */
            }
            if (firstChunk) {
                firstChunk = 0;
                uint8_t header[4 + MAX_FRAME_HEADER_SIZE];
                st32(header, MAGIC_NUMBER);
                int32_t h = 4 + write_frame_header(header + 4, closing ? chunkSize : -1, windowSize);
                CHECK_ARGUMENT(outCap - output >= h);
                memcpy(out + output, header, (size_t)h);
                output += h;
            }
            do {
                int32_t blockSize = chunkSize < blockSizeMax ? chunkSize : blockSizeMax;
                int lastBlock = closing && blockSize == chunkSize;
                const uint8_t* in = src + origin; /* the buffer */
                int32_t compressedSize = 0;       /* writeCompressedBlock :181-204 into compressed[0 .. compressedLength) */
                if (blockSize > 0) {
                    compressedSize = compress_block(c, in, offset, blockSize, compressed, SIZE_OF_BLOCK_HEADER, compressedLength - SIZE_OF_BLOCK_HEADER);
                }
                if (compressedSize == 0) {
                    st24(compressed, (uint32_t)((lastBlock ? 1 : 0) | (RAW_BLOCK << 1) | (blockSize << 3)));
                    if (blockSize > 0) {
                        memcpy(compressed + SIZE_OF_BLOCK_HEADER, in + offset, (size_t)blockSize);
                    }
                    compressedSize = SIZE_OF_BLOCK_HEADER + blockSize;
                }
                else {
                    st24(compressed, (uint32_t)((lastBlock ? 1 : 0) | (COMPRESSED_BLOCK << 1) | (compressedSize << 3)));
                    compressedSize += SIZE_OF_BLOCK_HEADER;
                }
                CHECK_ARGUMENT(outCap - output >= compressedSize); /* (the sink: the caller's buffer) */
                memcpy(out + output, compressed, (size_t)compressedSize);
                output += compressedSize;
                offset += blockSize;
                chunkSize -= blockSize;
            }
            while (chunkSize > 0);
            if (closing) {
                CHECK_ARGUMENT(outCap - output >= 4);
                st32(out + output, (uint32_t)orc_xxh64(src, n, 0)); /* (partialHash saw every chunk, in order) */
                output += 4;
                break;
            }
            /* slide :212-219 */
            int32_t slide = offset - windowSize;
            for (int32_t i = 0; i < (1 << c->p.hashLog); i++) {
                int32_t v = c->hashTable[i] - slide;
                c->hashTable[i] = v & ~(v >> 31);
            }
            for (int32_t i = 0; i < (1 << c->p.chainLog); i++) {
                int32_t v = c->chainTable[i] - slide;
                c->chainTable[i] = v & ~(v >> 31);
            }
            origin += slide;
            offset -= slide;
            position -= slide;
        }
    }
    return output;
}

static __thread cctx* g_cctx;
static void ensure_cctx(void);

int64_t orc_zstd_stream_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap)
{
    if (in_len < 0 || in_len >= (1LL << 30)) {
        return ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED);
    }
    ensure_cctx();
    fail_ctx f;
    f.status = 0;
    g_fail = &f;
    if (setjmp(f.jb)) {
        return f.status;
    }
    return zstd_stream_compress(g_cctx, in, (int32_t)in_len, out, out_cap);
}
int64_t orc_zstd_stream_max_compressed_length(int64_t n) { return orc_zstd_max_compressed_length(n) + 16; }

static void ensure_cctx(void)
{
    if (!g_cctx) {
        cctx* c = (cctx*)calloc(1, sizeof(cctx));
        c->hashTable = (int32_t*)malloc(sizeof(int32_t) << 17);
        c->chainTable = (int32_t*)malloc(sizeof(int32_t) << 16);
        int32_t maxSequences = MAX_BLOCK_SIZE / 4;
        c->ss.literalsBuffer = (uint8_t*)malloc(MAX_BLOCK_SIZE + 16);
        c->ss.offsets = (int32_t*)malloc(sizeof(int32_t) * maxSequences);
        c->ss.literalLengths = (int32_t*)malloc(sizeof(int32_t) * maxSequences);
        c->ss.matchLengths = (int32_t*)malloc(sizeof(int32_t) * maxSequences);
        c->ss.literalLengthCodes = (uint8_t*)malloc(maxSequences);
        c->ss.matchLengthCodes = (uint8_t*)malloc(maxSequences);
        c->ss.offsetCodes = (uint8_t*)malloc(maxSequences);
        g_cctx = c;
    }
}

int64_t orc_zstd_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap)
{
    ensure_cctx();
    /* The Java encoder writes with 8-byte stores that may run past the final size but never past the caller's
     * buffer unless it first fails a checkArgument; the restatement works in a scratch copy with slack so that
     * those transient stores cannot touch memory past out_cap, then copies the result back. */
    static __thread uint8_t* scratch;
    static __thread int64_t scratch_cap;
    if (scratch_cap < out_cap + 64) {
        free(scratch);
        scratch = (uint8_t*)malloc((size_t)out_cap + 64);
        scratch_cap = out_cap + 64;
    }
    fail_ctx f;
    f.status = 0;
    g_fail = &f;
    if (setjmp(f.jb)) {
        return f.status;
    }
    int64_t r = zstd_compress(g_cctx, in, (int32_t)in_len, scratch, out_cap);
    if (r > 0) {
        memcpy(out, scratch, (size_t)r);
    }
    return r;
}

/* frees the calling thread's encoder context (the timing driver's threads are short-lived) */
void orc_zstd_enc_thread_free(void)
{
    if (g_cctx) {
        cctx* c = g_cctx;
        free(c->hashTable);
        free(c->chainTable);
        free(c->ss.literalsBuffer);
        free(c->ss.offsets);
        free(c->ss.literalLengths);
        free(c->ss.matchLengths);
        free(c->ss.literalLengthCodes);
        free(c->ss.matchLengthCodes);
        free(c->ss.offsetCodes);
        free(c);
        g_cctx = 0;
    }
}
