/*
 * oracle/zstd_dec.c -- CPU restatement of the reference's Java Zstd frame decoder.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   frames/blocks ......... M/zstd/ZstdFrameDecompressor.java:135-310,860-962
 *   literals .............. M/zstd/ZstdFrameDecompressor.java:708-858, M/zstd/Huffman.java:52-324
 *   FSE tables ............ M/zstd/FseTableReader.java:27-168, M/zstd/FseCompressionTable.java:133-154
 *   FSE stream (weights) .. M/zstd/FiniteStateEntropy.java:38-151
 *   bit stream ............ M/zstd/BitInputStream.java:28-206
 *   sequences ............. M/zstd/ZstdFrameDecompressor.java:312-516,518-607,678-706
 *
 * Error offsets: the Java code passes absolute Unsafe addresses (heap base offset
 * included) to MalformedInputException; here they are reported relative to the start
 * of the input buffer.  Only the reason strings are asserted by the reference's tests
 * (T/zstd/AbstractTestZstd.java:69-78,175-184).
 * Java's unchecked exceptions on corrupt input (ArrayIndexOutOfBounds for weights > 12,
 * RLE symbols >= 128, 256 direct weights) are reported as "Input is corrupted".
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <stdlib.h>
#include <string.h>

#define SIZE_OF_LONG 8
#define MAGIC_NUMBER 0xFD2FB528u
#define V07_MAGIC_NUMBER 0xFD2FB527u
#define MIN_WINDOW_LOG 10
#define MAX_WINDOW_SIZE (1 << 23)
#define MAX_BLOCK_SIZE (128 * 1024)
#define MIN_BLOCK_SIZE 3
#define LONG_NUMBER_OF_SEQUENCES 0x7F00
#define MAX_LITERALS_LENGTH_SYMBOL 35
#define MAX_MATCH_LENGTH_SYMBOL 52
#define DEFAULT_MAX_OFFSET_CODE_SYMBOL 28
#define LITERAL_LENGTH_TABLE_LOG 9
#define MATCH_LENGTH_TABLE_LOG 9
#define OFFSET_TABLE_LOG 8
#define HUF_MAX_TABLE_LOG 12
#define HUF_MAX_FSE_TABLE_LOG 6
#define FSE_MAX_SYMBOL 255
#define FSE_MIN_TABLE_LOG 5

static const int32_t LITERALS_LENGTH_BASE[36] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
    16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000,
    0x2000, 0x4000, 0x8000, 0x10000};
static const int32_t MATCH_LENGTH_BASE[53] = {
    3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18,
    19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
    35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 0x83, 0x103, 0x203, 0x403, 0x803,
    0x1003, 0x2003, 0x4003, 0x8003, 0x10003};
static const int32_t OFFSET_CODES_BASE[29] = {
    0, 1, 1, 5, 0xD, 0x1D, 0x3D, 0x7D,
    0xFD, 0x1FD, 0x3FD, 0x7FD, 0xFFD, 0x1FFD, 0x3FFD, 0x7FFD,
    0xFFFD, 0x1FFFD, 0x3FFFD, 0x7FFFD, 0xFFFFD, 0x1FFFFD, 0x3FFFFD, 0x7FFFFD,
    0xFFFFFD, 0x1FFFFFD, 0x3FFFFFD, 0x7FFFFFD, 0xFFFFFFD};
static const uint8_t LITERALS_LENGTH_BITS[36] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12,
    13, 14, 15, 16};
static const uint8_t MATCH_LENGTH_BITS[53] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
    12, 13, 14, 15, 16};

typedef struct {
    int32_t log2_size;
    int32_t new_state[512];
    uint8_t symbol[512];
    uint8_t number_of_bits[512];
} fse_table;

/* Predefined distributions (RFC 8878 3.1.1.3.2.2); building them with the reference's own
 * table builder reproduces M/zstd/ZstdFrameDecompressor.java:85-113 (checked in tests). */
static const int16_t DEFAULT_LL_NORM[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t DEFAULT_OF_NORM[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const int16_t DEFAULT_ML_NORM[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};

typedef struct {
    const uint8_t* in;   /* whole input of the call */
    int64_t in_len;
    uint8_t* out;        /* whole output of the call */
    int64_t out_cap;
    int64_t err_off;
    int32_t err_detail;

    uint8_t literals[MAX_BLOCK_SIZE + SIZE_OF_LONG];
    const uint8_t* lit_ptr; /* current literals */
    int64_t lit_size;

    int32_t previous_offsets[3];
    fse_table ll_table, of_table, ml_table;
    const fse_table* cur_ll;
    const fse_table* cur_of;
    const fse_table* cur_ml;
    fse_table default_ll, default_of, default_ml;
    int defaults_built;

    /* Huffman -- M/zstd/Huffman.java:33-45 */
    int32_t huf_table_log;
    uint8_t huf_symbols[1 << HUF_MAX_TABLE_LOG];
    uint8_t huf_nbits[1 << HUF_MAX_TABLE_LOG];
    fse_table huf_fse;
} zctx;

#define FAILZ(c, detail, off)          \
    do {                               \
        (c)->err_detail = (detail);    \
        (c)->err_off = (off);          \
        return -1;                     \
    } while (0)
#define VERIFY(c, cond, detail, off)   \
    do {                               \
        if (!(cond)) FAILZ(c, detail, off); \
    } while (0)

/* bounds-guarded little-endian loads: bytes past the end of the whole input read as 0
 * (the Java code reads whatever follows in the array; valid streams never depend on it) */
static inline uint64_t rd_le(const zctx* c, int64_t pos, int n)
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) {
        int64_t p = pos + i;
        if (p >= 0 && p < c->in_len) v |= (uint64_t)c->in[p] << (8 * i);
    }
    return v;
}
static inline int32_t highest_bit(uint32_t v) { return 31 - __builtin_clz(v); }

/* ---- bit stream : BitInputStream.java ---- */
typedef struct {
    int64_t start, current;
    uint64_t bits;
    int32_t bits_consumed;
    int overflow;
} bitstream;

static inline uint64_t peek_bits(int32_t bits_consumed, uint64_t bits, int32_t n) /* :64-67 */
{
    return ((bits << (bits_consumed & 63)) >> 1) >> ((63 - n) & 63);
}
static inline uint64_t peek_bits_fast(int32_t bits_consumed, uint64_t bits, int32_t n) /* :74-77 */
{
    return (bits << (bits_consumed & 63)) >> ((64 - n) & 63);
}

/* Initializer.initialize :110-130 */
static int bit_init(zctx* c, bitstream* b, int64_t start, int64_t end)
{
    VERIFY(c, end - start >= 1, ACHIP_D_ZSTD_BITSTREAM_EMPTY, start);
    int32_t last_byte = (int32_t)rd_le(c, end - 1, 1);
    VERIFY(c, last_byte != 0, ACHIP_D_ZSTD_BITSTREAM_NO_MARK, end);
    b->start = start;
    b->overflow = 0;
    b->bits_consumed = SIZE_OF_LONG - highest_bit((uint32_t)last_byte);
    int32_t input_size = (int32_t)(end - start);
    if (input_size >= SIZE_OF_LONG) {
        b->current = end - SIZE_OF_LONG;
        b->bits = rd_le(c, b->current, 8);
    }
    else {
        b->current = start;
        b->bits = rd_le(c, start, input_size); /* readTail :39-59 */
        b->bits_consumed += (SIZE_OF_LONG - input_size) * 8;
    }
    return 0;
}

/* Loader.load :171-204; returns the Java method's boolean */
static int bit_load(const zctx* c, bitstream* b)
{
    if (b->bits_consumed > 64) {
        b->overflow = 1;
        return 1;
    }
    else if (b->current == b->start) {
        return 1;
    }
    int32_t bytes = (int32_t)((uint32_t)b->bits_consumed >> 3);
    if (b->current >= b->start + SIZE_OF_LONG) {
        if (bytes > 0) {
            b->current -= bytes;
            b->bits = rd_le(c, b->current, 8);
        }
        b->bits_consumed &= 7;
    }
    else if (b->current - bytes < b->start) {
        bytes = (int32_t)(b->current - b->start);
        b->current = b->start;
        b->bits_consumed -= bytes * SIZE_OF_LONG;
        b->bits = rd_le(c, b->start, 8);
        return 1;
    }
    else {
        b->current -= bytes;
        b->bits_consumed -= bytes * SIZE_OF_LONG;
        b->bits = rd_le(c, b->current, 8);
    }
    return 0;
}

/* ---- FSE decoding tables ---- */
/* shared tail of FseTableReader.readFseTable :127-159 + FseCompressionTable.spreadSymbols :138-154 */
static int fse_build(zctx* c, fse_table* table, const int16_t* norm, int32_t max_symbol, int32_t table_log, int64_t off)
{
    int16_t next_symbol[FSE_MAX_SYMBOL + 1];
    int32_t symbol_count = max_symbol + 1;
    int32_t table_size = 1 << table_log;
    int32_t high_threshold = table_size - 1;
    table->log2_size = table_log;
    for (int32_t s = 0; s < symbol_count; s++) {
        if (norm[s] == -1) {
            table->symbol[high_threshold--] = (uint8_t)s;
            next_symbol[s] = 1;
        }
        else {
            next_symbol[s] = norm[s];
        }
    }
    int32_t mask = table_size - 1;
    int32_t step = (table_size >> 1) + (table_size >> 3) + 3;
    int32_t position = 0;
    for (int32_t s = 0; s <= max_symbol; s++) {
        for (int32_t i = 0; i < norm[s]; i++) {
            table->symbol[position] = (uint8_t)s;
            do {
                position = (position + step) & mask;
            }
            while (position > high_threshold);
        }
    }
    VERIFY(c, position == 0, ACHIP_D_ZSTD_CORRUPTED, off);
    for (int32_t i = 0; i < table_size; i++) {
        uint8_t symbol = table->symbol[i];
        int16_t next_state = next_symbol[symbol]++;
        table->number_of_bits[i] = (uint8_t)(table_log - highest_bit((uint32_t)(uint16_t)next_state));
        table->new_state[i] = (int16_t)(((int32_t)next_state << table->number_of_bits[i]) - table_size);
    }
    return 0;
}

/* FseTableReader.readFseTable :27-160; returns bytes consumed or -1 */
static int64_t read_fse_table(zctx* c, fse_table* table, int64_t input_address, int64_t input_limit, int32_t max_symbol, int32_t max_table_log)
{
    int16_t norm[FSE_MAX_SYMBOL + 2];
    int64_t input = input_address;
    VERIFY(c, input_limit - input_address >= 4, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);

    int32_t threshold;
    int32_t symbol_number = 0;
    int previous_is_zero = 0;
    uint32_t bit_stream = (uint32_t)rd_le(c, input, 4);
    int32_t table_log = (int32_t)(bit_stream & 0xF) + FSE_MIN_TABLE_LOG;
    int32_t number_of_bits = table_log + 1;
    bit_stream >>= 4;
    int32_t bit_count = 4;
    VERIFY(c, table_log <= max_table_log, ACHIP_D_ZSTD_FSE_TABLE_LOG, input);

    int32_t remaining = (1 << table_log) + 1;
    threshold = 1 << table_log;

    while (remaining > 1 && symbol_number <= max_symbol) {
        if (previous_is_zero) {
            int32_t n0 = symbol_number;
            while ((bit_stream & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (input < input_limit - 5) {
                    input += 2;
                    bit_stream = (uint32_t)rd_le(c, input, 4) >> (bit_count & 31);
                }
                else {
                    bit_stream >>= 16;
                    bit_count += 16;
                }
            }
            while ((bit_stream & 3) == 3) {
                n0 += 3;
                bit_stream >>= 2;
                bit_count += 2;
            }
            n0 += (int32_t)(bit_stream & 3);
            bit_count += 2;
            VERIFY(c, n0 <= max_symbol, ACHIP_D_ZSTD_FSE_SYMBOL, input);
            while (symbol_number < n0) {
                norm[symbol_number++] = 0;
            }
            if ((input <= input_limit - 7) || (input + (bit_count >> 3) <= input_limit - 4)) {
                input += bit_count >> 3;
                bit_count &= 7;
                bit_stream = (uint32_t)rd_le(c, input, 4) >> (bit_count & 31);
            }
            else {
                bit_stream >>= 2;
            }
        }

        int16_t max = (int16_t)((2 * threshold - 1) - remaining);
        int16_t count;
        if ((int32_t)(bit_stream & (uint32_t)(threshold - 1)) < max) {
            count = (int16_t)(bit_stream & (uint32_t)(threshold - 1));
            bit_count += number_of_bits - 1;
        }
        else {
            count = (int16_t)(bit_stream & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) {
                count = (int16_t)(count - max);
            }
            bit_count += number_of_bits;
        }
        count--;

        remaining -= (count < 0) ? -count : count;
        norm[symbol_number++] = count;
        previous_is_zero = (count == 0);
        while (remaining < threshold) {
            number_of_bits--;
            threshold >>= 1;
        }

        if ((input <= input_limit - 7) || (input + (bit_count >> 3) <= input_limit - 4)) {
            input += bit_count >> 3;
            bit_count &= 7;
        }
        else {
            bit_count -= (int32_t)(8 * (input_limit - 4 - input));
            input = input_limit - 4;
        }
        bit_stream = (uint32_t)rd_le(c, input, 4) >> (bit_count & 31);
    }

    VERIFY(c, remaining == 1 && bit_count <= 32, ACHIP_D_ZSTD_CORRUPTED, input);
    max_symbol = symbol_number - 1;
    VERIFY(c, max_symbol <= FSE_MAX_SYMBOL, ACHIP_D_ZSTD_FSE_SYMBOL, input);
    input += (bit_count + 7) >> 3;

    if (fse_build(c, table, norm, max_symbol, table_log, input) < 0) {
        return -1;
    }
    return input - input_address;
}

/* FseTableReader.initializeRleTable :162-168 */
static void init_rle_table(fse_table* t, uint8_t value)
{
    t->log2_size = 0;
    t->symbol[0] = value;
    t->new_state[0] = 0;
    t->number_of_bits[0] = 0;
}

/* FiniteStateEntropy.decompress :38-151 (Huffman weight stream); returns count or -1 */
static int32_t fse_decompress(zctx* c, const fse_table* table, int64_t input_address, int64_t input_limit, uint8_t* out, int32_t out_len)
{
    int64_t input = input_address;
    int32_t output = 0;
    const int32_t output_limit = out_len;
    bitstream b;
    if (bit_init(c, &b, input, input_limit) < 0) return -1;

    int32_t state1 = (int32_t)peek_bits(b.bits_consumed, b.bits, table->log2_size);
    b.bits_consumed += table->log2_size;
    bit_load(c, &b);
    int32_t state2 = (int32_t)peek_bits(b.bits_consumed, b.bits, table->log2_size);
    b.bits_consumed += table->log2_size;
    bit_load(c, &b);

#define FSE_STEP(state)                                                                              \
    do {                                                                                             \
        int32_t nb_ = table->number_of_bits[state];                                                  \
        state = (int32_t)(table->new_state[state] + (int32_t)peek_bits(b.bits_consumed, b.bits, nb_)); \
        b.bits_consumed += nb_;                                                                      \
    } while (0)

    while (output <= output_limit - 4) {
        out[output] = table->symbol[state1];
        FSE_STEP(state1);
        out[output + 1] = table->symbol[state2];
        FSE_STEP(state2);
        out[output + 2] = table->symbol[state1];
        FSE_STEP(state1);
        out[output + 3] = table->symbol[state2];
        FSE_STEP(state2);
        output += 4;
        if (bit_load(c, &b)) {
            break;
        }
    }

    for (;;) {
        VERIFY(c, output <= output_limit - 2, ACHIP_D_ZSTD_FSE_OUTPUT_SMALL, input);
        out[output++] = table->symbol[state1];
        FSE_STEP(state1);
        b.overflow = 0;
        bit_load(c, &b);
        if (b.overflow) {
            out[output++] = table->symbol[state2];
            break;
        }
        VERIFY(c, output <= output_limit - 2, ACHIP_D_ZSTD_FSE_OUTPUT_SMALL, input);
        out[output++] = table->symbol[state2];
        FSE_STEP(state2);
        b.overflow = 0;
        bit_load(c, &b);
        if (b.overflow) {
            out[output++] = table->symbol[state1];
            break;
        }
    }
#undef FSE_STEP
    return output;
}

/* ---- Huffman : Huffman.java ---- */
/* readTable :52-128 ; returns bytes consumed or -1 */
static int32_t huf_read_table(zctx* c, int64_t input_address, int32_t size)
{
    uint8_t weights[FSE_MAX_SYMBOL + 2];
    int32_t ranks[HUF_MAX_TABLE_LOG + 1];
    memset(ranks, 0, sizeof(ranks));
    memset(weights, 0, sizeof(weights));
    int64_t input = input_address;

    VERIFY(c, size > 0, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    int32_t input_size = (int32_t)rd_le(c, input++, 1);
    int32_t output_size;
    if (input_size >= 128) {
        output_size = input_size - 127;
        input_size = (output_size + 1) / 2;
        VERIFY(c, input_size + 1 <= size, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        VERIFY(c, output_size <= FSE_MAX_SYMBOL + 1, ACHIP_D_ZSTD_CORRUPTED, input);
        for (int32_t i = 0; i < output_size; i += 2) {
            int32_t value = (int32_t)rd_le(c, input + i / 2, 1);
            weights[i] = (uint8_t)(value >> 4);
            weights[i + 1] = (uint8_t)(value & 0xF);
        }
    }
    else {
        VERIFY(c, input_size + 1 <= size, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        int64_t input_limit = input + input_size;
        int64_t n = read_fse_table(c, &c->huf_fse, input, input_limit, FSE_MAX_SYMBOL, HUF_MAX_FSE_TABLE_LOG);
        if (n < 0) return -1;
        input += n;
        output_size = fse_decompress(c, &c->huf_fse, input, input_limit, weights, FSE_MAX_SYMBOL + 1);
        if (output_size < 0) return -1;
    }

    int32_t total_weight = 0;
    for (int32_t i = 0; i < output_size; i++) {
        VERIFY(c, weights[i] <= HUF_MAX_TABLE_LOG, ACHIP_D_ZSTD_CORRUPTED, input); /* Java: ArrayIndexOutOfBounds */
        ranks[weights[i]]++;
        total_weight += (1 << weights[i]) >> 1;
    }
    VERIFY(c, total_weight != 0, ACHIP_D_ZSTD_CORRUPTED, input);
    int32_t table_log = highest_bit((uint32_t)total_weight) + 1;
    VERIFY(c, table_log <= HUF_MAX_TABLE_LOG, ACHIP_D_ZSTD_CORRUPTED, input);
    int32_t total = 1 << table_log;
    int32_t rest = total - total_weight;
    VERIFY(c, (rest & (rest - 1)) == 0, ACHIP_D_ZSTD_CORRUPTED, input);
    int32_t last_weight = highest_bit((uint32_t)rest) + 1;
    VERIFY(c, output_size <= FSE_MAX_SYMBOL, ACHIP_D_ZSTD_CORRUPTED, input); /* Java: weights[256] out of bounds */
    weights[output_size] = (uint8_t)last_weight;
    ranks[last_weight]++;
    int32_t number_of_symbols = output_size + 1;

    int32_t next_rank_start = 0;
    for (int32_t i = 1; i < table_log + 1; ++i) {
        int32_t current = next_rank_start;
        next_rank_start += ranks[i] << (i - 1);
        ranks[i] = current;
    }
    for (int32_t n = 0; n < number_of_symbols; n++) {
        int32_t weight = weights[n];
        int32_t length = (1 << weight) >> 1;
        uint8_t nbits = (uint8_t)(table_log + 1 - weight);
        for (int32_t i = ranks[weight]; i < ranks[weight] + length; i++) {
            c->huf_symbols[i] = (uint8_t)n;
            c->huf_nbits[i] = nbits;
        }
        ranks[weight] += length;
    }
    VERIFY(c, ranks[1] >= 2 && (ranks[1] & 1) == 0, ACHIP_D_ZSTD_CORRUPTED, input);
    c->huf_table_log = table_log;
    return input_size + 1;
}

static inline int32_t huf_decode_symbol(const zctx* c, uint8_t* out, int64_t pos, uint64_t bits, int32_t bits_consumed)
{ /* Huffman.decodeSymbol :319-324 */
    int32_t value = (int32_t)peek_bits_fast(bits_consumed, bits, c->huf_table_log);
    out[pos] = c->huf_symbols[value];
    return bits_consumed + c->huf_nbits[value];
}

/* Huffman.decodeTail :291-317 */
static int huf_decode_tail(zctx* c, bitstream* b, uint8_t* out, int64_t output, int64_t output_limit)
{
    while (output < output_limit) {
        if (bit_load(c, b)) {
            break;
        }
        b->bits_consumed = huf_decode_symbol(c, out, output++, b->bits, b->bits_consumed);
    }
    while (output < output_limit) {
        b->bits_consumed = huf_decode_symbol(c, out, output++, b->bits, b->bits_consumed);
    }
    VERIFY(c, b->start == b->current && b->bits_consumed == 64, ACHIP_D_ZSTD_BITSTREAM_NOT_CONSUMED, b->start);
    return 0;
}

/* Huffman.decodeSingleStream :130-164 */
static int huf_decode_single(zctx* c, int64_t input_address, int64_t input_limit, uint8_t* out, int64_t output_limit)
{
    bitstream b;
    if (bit_init(c, &b, input_address, input_limit) < 0) return -1;
    int64_t output = 0;
    int64_t fast_output_limit = output_limit - 4;
    while (output < fast_output_limit) {
        if (bit_load(c, &b)) {
            break;
        }
        b.bits_consumed = huf_decode_symbol(c, out, output, b.bits, b.bits_consumed);
        b.bits_consumed = huf_decode_symbol(c, out, output + 1, b.bits, b.bits_consumed);
        b.bits_consumed = huf_decode_symbol(c, out, output + 2, b.bits, b.bits_consumed);
        b.bits_consumed = huf_decode_symbol(c, out, output + 3, b.bits, b.bits_consumed);
        output += 4;
    }
    return huf_decode_tail(c, &b, out, output, output_limit);
}

/* Huffman.decode4Streams :166-289 */
static int huf_decode_4streams(zctx* c, int64_t input_address, int64_t input_limit, uint8_t* out, int64_t output_limit)
{
    VERIFY(c, input_limit - input_address >= 10, ACHIP_D_ZSTD_CORRUPTED, input_address);
    int64_t start1 = input_address + 6;
    int64_t start2 = start1 + (int64_t)rd_le(c, input_address, 2);
    int64_t start3 = start2 + (int64_t)rd_le(c, input_address + 2, 2);
    int64_t start4 = start3 + (int64_t)rd_le(c, input_address + 4, 2);
    VERIFY(c, start2 < start3 && start3 < start4 && start4 < input_limit, ACHIP_D_ZSTD_CORRUPTED, input_address);

    bitstream s1, s2, s3, s4;
    if (bit_init(c, &s1, start1, start2) < 0) return -1;
    if (bit_init(c, &s2, start2, start3) < 0) return -1;
    if (bit_init(c, &s3, start3, start4) < 0) return -1;
    if (bit_init(c, &s4, start4, input_limit) < 0) return -1;

    int32_t segment_size = (int32_t)((output_limit + 3) / 4);
    int64_t output_start2 = segment_size;
    int64_t output_start3 = output_start2 + segment_size;
    int64_t output_start4 = output_start3 + segment_size;
    int64_t o1 = 0, o2 = output_start2, o3 = output_start3, o4 = output_start4;
    int64_t fast_output_limit = output_limit - 7;

    while (o4 < fast_output_limit) {
        for (int k = 0; k < 4; k++) {
            s1.bits_consumed = huf_decode_symbol(c, out, o1 + k, s1.bits, s1.bits_consumed);
            s2.bits_consumed = huf_decode_symbol(c, out, o2 + k, s2.bits, s2.bits_consumed);
            s3.bits_consumed = huf_decode_symbol(c, out, o3 + k, s3.bits, s3.bits_consumed);
            s4.bits_consumed = huf_decode_symbol(c, out, o4 + k, s4.bits, s4.bits_consumed);
        }
        o1 += 4;
        o2 += 4;
        o3 += 4;
        o4 += 4;
        if (bit_load(c, &s1)) break;
        if (bit_load(c, &s2)) break;
        if (bit_load(c, &s3)) break;
        if (bit_load(c, &s4)) break;
    }
    VERIFY(c, o1 <= output_start2 && o2 <= output_start3 && o3 <= output_start4, ACHIP_D_ZSTD_CORRUPTED, input_address);

    if (huf_decode_tail(c, &s1, out, o1, output_start2) < 0) return -1;
    if (huf_decode_tail(c, &s2, out, o2, output_start3) < 0) return -1;
    if (huf_decode_tail(c, &s3, out, o3, output_start4) < 0) return -1;
    if (huf_decode_tail(c, &s4, out, o4, output_limit) < 0) return -1;
    return 0;
}

/* ---- literals sections ---- */
/* decodeCompressedLiterals :708-774 ; returns bytes consumed or -1 */
static int32_t decode_compressed_literals(zctx* c, int64_t input_address, int32_t block_size, int32_t literals_block_type)
{
    int64_t input = input_address;
    VERIFY(c, block_size >= 5, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    int32_t compressed_size, uncompressed_size, header_size;
    int single_stream = 0;
    int32_t type = ((int32_t)rd_le(c, input, 1) >> 2) & 3;
    switch (type) {
        case 0:
            single_stream = 1; /* fallthrough */
        case 1: {
            uint32_t header = (uint32_t)rd_le(c, input, 4);
            header_size = 3;
            uncompressed_size = (int32_t)((header >> 4) & 0x3FF);
            compressed_size = (int32_t)((header >> 14) & 0x3FF);
            break;
        }
        case 2: {
            uint32_t header = (uint32_t)rd_le(c, input, 4);
            header_size = 4;
            uncompressed_size = (int32_t)((header >> 4) & 0x3FFF);
            compressed_size = (int32_t)((header >> 18) & 0x3FFF);
            break;
        }
        default: {
            uint64_t header = rd_le(c, input, 5);
            header_size = 5;
            uncompressed_size = (int32_t)((header >> 4) & 0x3FFFF);
            compressed_size = (int32_t)((header >> 22) & 0x3FFFF);
            break;
        }
    }
    VERIFY(c, uncompressed_size <= MAX_BLOCK_SIZE, ACHIP_D_ZSTD_LITERALS_TOO_LARGE, input);
    VERIFY(c, header_size + compressed_size <= block_size, ACHIP_D_ZSTD_CORRUPTED, input);
    input += header_size;
    int64_t input_limit = input + compressed_size;
    if (literals_block_type != 3) {
        int32_t n = huf_read_table(c, input, compressed_size);
        if (n < 0) return -1;
        input += n;
    }
    c->lit_ptr = c->literals;
    c->lit_size = uncompressed_size;
    if (single_stream) {
        if (huf_decode_single(c, input, input_limit, c->literals, uncompressed_size) < 0) return -1;
    }
    else {
        if (huf_decode_4streams(c, input, input_limit, c->literals, uncompressed_size) < 0) return -1;
    }
    return header_size + compressed_size;
}

/* decodeRleLiterals :776-810 */
static int32_t decode_rle_literals(zctx* c, int64_t input_address, int32_t block_size)
{
    int64_t input = input_address;
    int32_t output_size;
    int32_t type = ((int32_t)rd_le(c, input, 1) >> 2) & 3;
    switch (type) {
        case 0:
        case 2:
            output_size = (int32_t)rd_le(c, input, 1) >> 3;
            input++;
            break;
        case 1:
            output_size = (int32_t)rd_le(c, input, 2) >> 4;
            input += 2;
            break;
        default:
            VERIFY(c, block_size >= 4, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            output_size = (int32_t)(rd_le(c, input, 4) & 0xFFFFFF) >> 4;
            input += 3;
            break;
    }
    VERIFY(c, output_size <= MAX_BLOCK_SIZE, ACHIP_D_ZSTD_LITERALS_TOO_LARGE, input);
    uint8_t value = (uint8_t)rd_le(c, input++, 1);
    memset(c->literals, value, (size_t)output_size + SIZE_OF_LONG);
    c->lit_ptr = c->literals;
    c->lit_size = output_size;
    return (int32_t)(input - input_address);
}

/* decodeRawLiterals :812-858 (the in-place / copied distinction is a CPU over-read guard only) */
static int32_t decode_raw_literals(zctx* c, int64_t input_address, int64_t input_limit)
{
    int64_t input = input_address;
    int32_t type = ((int32_t)rd_le(c, input, 1) >> 2) & 3;
    int32_t literal_size;
    switch (type) {
        case 0:
        case 2:
            literal_size = (int32_t)rd_le(c, input, 1) >> 3;
            input++;
            break;
        case 1:
            literal_size = (int32_t)rd_le(c, input, 2) >> 4;
            input += 2;
            break;
        default:
            literal_size = (int32_t)rd_le(c, input, 3) >> 4;
            input += 3;
            break;
    }
    VERIFY(c, input + literal_size <= input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    c->lit_ptr = c->in + input;
    c->lit_size = literal_size;
    input += literal_size;
    return (int32_t)(input - input_address);
}

/* ---- sequences ---- */
static void build_defaults(zctx* c)
{
    if (c->defaults_built) return;
    fse_build(c, &c->default_ll, DEFAULT_LL_NORM, 35, 6, 0);
    fse_build(c, &c->default_of, DEFAULT_OF_NORM, 28, 5, 0);
    fse_build(c, &c->default_ml, DEFAULT_ML_NORM, 52, 6, 0);
    c->defaults_built = 1;
}

/* computeLiteralsTable / computeOffsetsTable / computeMatchLengthTable :609-676 */
static int64_t compute_table(zctx* c, int32_t type, int64_t input, int64_t input_limit, fse_table* own, const fse_table* dflt,
                             const fse_table** cur, int32_t max_symbol, int32_t max_log)
{
    switch (type) {
        case 1: { /* RLE */
            VERIFY(c, input < input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            int8_t value = (int8_t)rd_le(c, input++, 1);
            VERIFY(c, value <= max_symbol, ACHIP_D_ZSTD_VALUE_TOO_LARGE, input);
            VERIFY(c, value >= 0, ACHIP_D_ZSTD_CORRUPTED, input); /* Java: negative array index later */
            init_rle_table(own, (uint8_t)value);
            *cur = own;
            break;
        }
        case 0:
            *cur = dflt;
            break;
        case 3:
            VERIFY(c, *cur != NULL, ACHIP_D_ZSTD_TABLE_MISSING, input);
            break;
        default: {
            int64_t n = read_fse_table(c, own, input, input_limit, max_symbol, max_log);
            if (n < 0) return -1;
            input += n;
            *cur = own;
            break;
        }
    }
    return input;
}

/* decompressSequences :312-516 ; returns decoded size or -1 */
static int64_t decompress_sequences(zctx* c, int64_t input_address, int64_t input_limit, int64_t output_address)
{
    const int64_t output_limit = c->out_cap;
    int64_t input = input_address;
    int64_t output = output_address;
    int64_t literals_input = 0;
    const int64_t literals_limit = c->lit_size;

    int32_t size = (int32_t)(input_limit - input_address);
    VERIFY(c, size >= 1, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);

    int32_t sequence_count = (int32_t)rd_le(c, input++, 1);
    if (sequence_count != 0) {
        if (sequence_count == 255) {
            VERIFY(c, input + 2 <= input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            sequence_count = (int32_t)rd_le(c, input, 2) + LONG_NUMBER_OF_SEQUENCES;
            input += 2;
        }
        else if (sequence_count > 127) {
            VERIFY(c, input < input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            sequence_count = ((sequence_count - 128) << 8) + (int32_t)rd_le(c, input++, 1);
        }
        VERIFY(c, input + 4 <= input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);

        int32_t type = (int32_t)rd_le(c, input++, 1);
        int32_t ll_type = type >> 6;
        int32_t of_type = (type >> 4) & 3;
        int32_t ml_type = (type >> 2) & 3;

        input = compute_table(c, ll_type, input, input_limit, &c->ll_table, &c->default_ll, &c->cur_ll, MAX_LITERALS_LENGTH_SYMBOL, LITERAL_LENGTH_TABLE_LOG);
        if (input < 0) return -1;
        input = compute_table(c, of_type, input, input_limit, &c->of_table, &c->default_of, &c->cur_of, DEFAULT_MAX_OFFSET_CODE_SYMBOL, OFFSET_TABLE_LOG);
        if (input < 0) return -1;
        input = compute_table(c, ml_type, input, input_limit, &c->ml_table, &c->default_ml, &c->cur_ml, MAX_MATCH_LENGTH_SYMBOL, MATCH_LENGTH_TABLE_LOG);
        if (input < 0) return -1;

        bitstream b;
        if (bit_init(c, &b, input, input_limit) < 0) return -1;
        const fse_table* llt = c->cur_ll;
        const fse_table* oft = c->cur_of;
        const fse_table* mlt = c->cur_ml;

        int32_t ll_state = (int32_t)peek_bits(b.bits_consumed, b.bits, llt->log2_size);
        b.bits_consumed += llt->log2_size;
        int32_t of_state = (int32_t)peek_bits(b.bits_consumed, b.bits, oft->log2_size);
        b.bits_consumed += oft->log2_size;
        int32_t ml_state = (int32_t)peek_bits(b.bits_consumed, b.bits, mlt->log2_size);
        b.bits_consumed += mlt->log2_size;

        int32_t* prev = c->previous_offsets;

        while (sequence_count > 0) {
            sequence_count--;
            b.overflow = 0;
            bit_load(c, &b);
            if (b.overflow) {
                VERIFY(c, sequence_count == 0, ACHIP_D_ZSTD_SEQUENCES_NOT_CONSUMED, input);
                break;
            }

            int32_t ll_code = llt->symbol[ll_state];
            int32_t ml_code = mlt->symbol[ml_state];
            int32_t of_code = oft->symbol[of_state];
            /* table builders bound the symbols (max_symbol arguments above) */
            int32_t ll_bits = LITERALS_LENGTH_BITS[ll_code];
            int32_t ml_bits = MATCH_LENGTH_BITS[ml_code];
            int32_t of_bits = of_code;

            int32_t offset = OFFSET_CODES_BASE[of_code];
            if (of_code > 0) {
                offset += (int32_t)peek_bits(b.bits_consumed, b.bits, of_bits);
                b.bits_consumed += of_bits;
            }

            if (of_code <= 1) {
                if (ll_code == 0) {
                    offset++;
                }
                if (offset != 0) {
                    int32_t temp;
                    if (offset == 3) {
                        temp = prev[0] - 1;
                    }
                    else {
                        temp = prev[offset];
                    }
                    if (temp == 0) {
                        temp = 1;
                    }
                    if (offset != 1) {
                        prev[2] = prev[1];
                    }
                    prev[1] = prev[0];
                    prev[0] = temp;
                    offset = temp;
                }
                else {
                    offset = prev[0];
                }
            }
            else {
                prev[2] = prev[1];
                prev[1] = prev[0];
                prev[0] = offset;
            }

            int32_t match_length = MATCH_LENGTH_BASE[ml_code];
            if (ml_code > 31) {
                match_length += (int32_t)peek_bits(b.bits_consumed, b.bits, ml_bits);
                b.bits_consumed += ml_bits;
            }
            int32_t literals_length = LITERALS_LENGTH_BASE[ll_code];
            if (ll_code > 15) {
                literals_length += (int32_t)peek_bits(b.bits_consumed, b.bits, ll_bits);
                b.bits_consumed += ll_bits;
            }

            int32_t total_bits = ll_bits + ml_bits + of_bits;
            if (total_bits > 64 - 7 - (LITERAL_LENGTH_TABLE_LOG + MATCH_LENGTH_TABLE_LOG + OFFSET_TABLE_LOG)) {
                bit_load(c, &b);
            }

            int32_t nb;
            nb = llt->number_of_bits[ll_state];
            ll_state = (int32_t)(llt->new_state[ll_state] + (int32_t)peek_bits(b.bits_consumed, b.bits, nb));
            b.bits_consumed += nb;
            nb = mlt->number_of_bits[ml_state];
            ml_state = (int32_t)(mlt->new_state[ml_state] + (int32_t)peek_bits(b.bits_consumed, b.bits, nb));
            b.bits_consumed += nb;
            nb = oft->number_of_bits[of_state];
            of_state = (int32_t)(oft->new_state[of_state] + (int32_t)peek_bits(b.bits_consumed, b.bits, nb));
            b.bits_consumed += nb;

            const int64_t literal_output_limit = output + literals_length;
            const int64_t match_output_limit = literal_output_limit + match_length;
            VERIFY(c, match_output_limit <= output_limit, ACHIP_D_ZSTD_OUTPUT_TOO_SMALL, input);
            int64_t literal_end = literals_input + literals_length;
            VERIFY(c, literal_end <= literals_limit, ACHIP_D_ZSTD_CORRUPTED, input);
            int64_t match_address = literal_output_limit - offset;
            VERIFY(c, match_address >= 0, ACHIP_D_ZSTD_CORRUPTED, input); /* >= start of the whole call's output :496 */

            memcpy(c->out + output, c->lit_ptr + literals_input, (size_t)literals_length);
            for (int64_t i = 0; i < match_length; i++) {
                c->out[literal_output_limit + i] = c->out[match_address + i];
            }
            output = match_output_limit;
            literals_input = literal_end;
        }
    }

    /* copyLastLiteral :518-525 */
    int64_t last_literals_size = literals_limit - literals_input;
    VERIFY(c, output + last_literals_size <= output_limit, ACHIP_D_ZSTD_OUTPUT_TOO_SMALL, input);
    memcpy(c->out + output, c->lit_ptr + literals_input, (size_t)last_literals_size);
    output += last_literals_size;
    return output - output_address;
}

/* decodeCompressedBlock :265-310 */
static int64_t decode_compressed_block(zctx* c, int64_t input_address, int32_t block_size, int64_t output_address, int32_t window_size)
{
    int64_t input_limit = input_address + block_size;
    int64_t input = input_address;
    VERIFY(c, block_size <= MAX_BLOCK_SIZE, ACHIP_D_ZSTD_BLOCK_TOO_LARGE, input);
    VERIFY(c, block_size >= MIN_BLOCK_SIZE, ACHIP_D_ZSTD_BLOCK_TOO_SMALL, input);

    int32_t literals_block_type = (int32_t)rd_le(c, input, 1) & 3;
    int32_t n;
    switch (literals_block_type) {
        case 0:
            n = decode_raw_literals(c, input, input_limit);
            break;
        case 1:
            n = decode_rle_literals(c, input, block_size);
            break;
        case 3:
            VERIFY(c, c->huf_table_log != -1, ACHIP_D_ZSTD_DICT_CORRUPTED, input);
            /* fallthrough */
        default:
            n = decode_compressed_literals(c, input, block_size, literals_block_type);
            break;
    }
    if (n < 0) return -1;
    input += n;
    VERIFY(c, window_size <= MAX_WINDOW_SIZE, ACHIP_D_ZSTD_WINDOW_TOO_LARGE, input);
    return decompress_sequences(c, input, input_address + block_size, output_address);
}

typedef struct {
    int64_t header_size;
    int32_t window_size;
    int64_t content_size;
    int has_checksum;
} frame_header;

/* readFrameHeader :860-940 */
static int read_frame_header(zctx* c, int64_t input_address, int64_t input_limit, frame_header* fh)
{
    int64_t input = input_address;
    VERIFY(c, input < input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    int32_t fhd = (int32_t)rd_le(c, input++, 1);
    int single_segment = (fhd & 0x20) != 0;
    int32_t dictionary_descriptor = fhd & 3;
    int32_t content_size_descriptor = fhd >> 6;
    int32_t header_size = 1 + (single_segment ? 0 : 1) + (dictionary_descriptor == 0 ? 0 : (1 << (dictionary_descriptor - 1))) +
                          (content_size_descriptor == 0 ? (single_segment ? 1 : 0) : (1 << content_size_descriptor));
    VERIFY(c, header_size <= input_limit - input_address, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);

    int32_t window_size = -1;
    if (!single_segment) {
        int32_t window_descriptor = (int32_t)rd_le(c, input++, 1);
        int32_t exponent = window_descriptor >> 3;
        int32_t mantissa = window_descriptor & 7;
        uint32_t base = 1u << ((MIN_WINDOW_LOG + exponent) & 31);
        window_size = (int32_t)(base + ((int32_t)base / 8) * mantissa);
    }
    int64_t dictionary_id = -1;
    switch (dictionary_descriptor) {
        case 1: dictionary_id = (int64_t)rd_le(c, input, 1); input += 1; break;
        case 2: dictionary_id = (int64_t)rd_le(c, input, 2); input += 2; break;
        case 3: dictionary_id = (int64_t)rd_le(c, input, 4); input += 4; break;
        default: break;
    }
    VERIFY(c, dictionary_id == -1, ACHIP_D_ZSTD_DICTIONARY, input);

    int64_t content_size = -1;
    switch (content_size_descriptor) {
        case 0:
            if (single_segment) {
                content_size = (int64_t)rd_le(c, input, 1);
                input += 1;
            }
            break;
        case 1: content_size = (int64_t)rd_le(c, input, 2) + 256; input += 2; break;
        case 2: content_size = (int64_t)rd_le(c, input, 4); input += 4; break;
        default: content_size = (int64_t)rd_le(c, input, 8); input += 8; break;
    }
    fh->header_size = input - input_address;
    fh->window_size = window_size;
    fh->content_size = content_size;
    fh->has_checksum = (fhd & 4) != 0;
    return 0;
}

/* verifyMagic :949-962 */
static int verify_magic(zctx* c, int64_t input_address, int64_t input_limit)
{
    VERIFY(c, input_limit - input_address >= 4, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input_address);
    uint32_t magic = (uint32_t)rd_le(c, input_address, 4);
    if (magic != MAGIC_NUMBER) {
        if (magic == V07_MAGIC_NUMBER) FAILZ(c, ACHIP_D_ZSTD_V07_MAGIC, input_address);
        FAILZ(c, ACHIP_D_ZSTD_BAD_MAGIC, input_address);
    }
    return 4;
}

/* ZstdFrameDecompressor.decompress :135-210 */
static int64_t zstd_decompress(zctx* c)
{
    if (c->out_cap == 0) {
        return 0;
    }
    const int64_t input_limit = c->in_len;
    int64_t input = 0;
    int64_t output = 0;
    build_defaults(c);
    c->huf_table_log = -1; /* the Huffman object lives as long as the decompressor; a fresh oracle call = a fresh decompressor */

    while (input < input_limit) {
        /* reset() :212-221 */
        c->previous_offsets[0] = 1;
        c->previous_offsets[1] = 4;
        c->previous_offsets[2] = 8;
        c->cur_ll = c->cur_of = c->cur_ml = NULL;
        int64_t output_start = output;

        if (verify_magic(c, input, input_limit) < 0) return -1;
        input += 4;
        frame_header fh;
        if (read_frame_header(c, input, input_limit, &fh) < 0) return -1;
        input += fh.header_size;

        int last_block;
        do {
            VERIFY(c, input + 3 <= input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            int32_t header = (int32_t)rd_le(c, input, 3);
            input += 3;
            last_block = (header & 1) != 0;
            int32_t block_type = (header >> 1) & 3;
            int32_t block_size = (header >> 3) & 0x1FFFFF;
            int64_t decoded_size;
            switch (block_type) {
                case 0: /* decodeRawBlock :223-229 */
                    VERIFY(c, input + block_size <= input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
                    VERIFY(c, output + block_size <= c->out_cap, ACHIP_D_ZSTD_OUTPUT_TOO_SMALL, input);
                    memcpy(c->out + output, c->in + input, (size_t)block_size);
                    decoded_size = block_size;
                    input += block_size;
                    break;
                case 1: /* decodeRleBlock :231-263 */
                    VERIFY(c, input + 1 <= input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
                    VERIFY(c, output + block_size <= c->out_cap, ACHIP_D_ZSTD_OUTPUT_TOO_SMALL, input);
                    memset(c->out + output, c->in[input], (size_t)block_size);
                    decoded_size = block_size;
                    input += 1;
                    break;
                case 2:
                    VERIFY(c, input + block_size <= input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
                    decoded_size = decode_compressed_block(c, input, block_size, output, fh.window_size);
                    if (decoded_size < 0) return -1;
                    input += block_size;
                    break;
                default:
                    FAILZ(c, ACHIP_D_ZSTD_INVALID_BLOCK_TYPE, input);
            }
            output += decoded_size;
        }
        while (!last_block);

        if (fh.has_checksum) {
            int64_t decoded_frame_size = output - output_start;
            uint64_t hash = orc_xxh64(c->out + output_start, decoded_frame_size, 0);
            VERIFY(c, input + 4 <= input_limit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            uint32_t checksum = (uint32_t)rd_le(c, input, 4);
            if (checksum != (uint32_t)hash) {
                FAILZ(c, ACHIP_D_ZSTD_BAD_CHECKSUM, input);
            }
            input += 4;
        }
    }
    return output;
}

static __thread zctx* g_ctx;

int64_t orc_zstd_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off)
{
    if (!g_ctx) {
        g_ctx = (zctx*)calloc(1, sizeof(zctx));
    }
    zctx* c = g_ctx;
    c->in = in;
    c->in_len = in_len;
    c->out = out;
    c->out_cap = out_cap;
    c->err_off = 0;
    c->err_detail = 0;
    int64_t r = zstd_decompress(c);
    if (r < 0) {
        if (err_off) *err_off = c->err_off;
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, c->err_detail);
    }
    return r;
}

/* ZstdFrameDecompressor.getDecompressedSize :942-947 */
int64_t orc_zstd_decompressed_size(const uint8_t* in, int64_t in_len, int64_t* err_off)
{
    if (!g_ctx) {
        g_ctx = (zctx*)calloc(1, sizeof(zctx));
    }
    zctx* c = g_ctx;
    c->in = in;
    c->in_len = in_len;
    frame_header fh;
    if (verify_magic(c, 0, in_len) < 0 || read_frame_header(c, 4, in_len, &fh) < 0) {
        if (err_off) *err_off = c->err_off;
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, c->err_detail);
    }
    return fh.content_size;
}

/* test hook: ZstdFrameDecompressor.readFrameHeader on a bare header (T/zstd/TestCompressor.java:100-110);
 * out4 = {headerSize, windowSize, contentSize, hasChecksum} */
int32_t orc_zstd_read_frame_header(const uint8_t* in, int64_t in_len, int64_t* out4)
{
    if (!g_ctx) {
        g_ctx = (zctx*)calloc(1, sizeof(zctx));
    }
    zctx* c = g_ctx;
    c->in = in;
    c->in_len = in_len;
    frame_header fh;
    if (read_frame_header(c, 0, in_len, &fh) < 0) {
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, c->err_detail);
    }
    out4[0] = fh.header_size;
    out4[1] = fh.window_size;
    out4[2] = fh.content_size;
    out4[3] = fh.has_checksum;
    return 0;
}

/* ZstdJavaCompressor.maxCompressedLength  M/zstd/ZstdJavaCompressor.java:31-40 */
int64_t orc_zstd_max_compressed_length(int64_t n)
{
    int64_t result = n + (int64_t)((uint32_t)n >> 8);
    if (n < MAX_BLOCK_SIZE) {
        result += (int64_t)((uint32_t)(MAX_BLOCK_SIZE - n) >> 11);
    }
    return result;
}

/* frees the calling thread's decoder context (the timing driver's threads are short-lived) */
void orc_zstd_dec_thread_free(void)
{
    free(g_ctx);
    g_ctx = 0;
}
