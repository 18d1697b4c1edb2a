/*
 * oracle/lz4.c -- CPU restatement of the reference's Java LZ4 block codec.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   compress   follows M/lz4/Lz4RawCompressor.java:69-312
 *   decompress follows M/lz4/Lz4RawDecompressor.java:35-198
 *
 * Java semantics kept: wrap-around 32/64-bit arithmetic, unaligned little-endian
 * loads, table entries = positions relative to the input start with 0 meaning
 * both "empty" and "position 0".
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <string.h>

static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint16_t ld16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

enum {
    HASH_LOG = 12,
    MIN_TABLE_SIZE = 16,
    MAX_TABLE_SIZE = 1 << HASH_LOG,
    COPY_LENGTH = 8,
    MIN_MATCH = 4,
    LAST_LITERAL_SIZE = 5,
    MATCH_FIND_LIMIT = COPY_LENGTH + MIN_MATCH,
    MIN_LENGTH = MATCH_FIND_LIMIT + 1,
    ML_BITS = 4,
    ML_MASK = (1 << ML_BITS) - 1,
    RUN_BITS = 8 - ML_BITS,
    RUN_MASK = (1 << RUN_BITS) - 1,
    MAX_DISTANCE = (1 << 16) - 1,
    SKIP_TRIGGER = 6
};
#define MAX_INPUT_SIZE 0x7E000000LL

/* Lz4RawCompressor.hash :50-62 */
static inline int32_t lz4_hash(uint64_t value, int32_t mask)
{
    return (int32_t)(((value * 889523592379ULL) >> 28) & (uint64_t)(uint32_t)mask);
}

/* Lz4RawCompressor.maxCompressedLength :64-67 */
int64_t orc_lz4_max_compressed_length(int64_t n) { return n + n / 255 + 16; }

/* Lz4RawCompressor.computeTableSize :304-311 */
static int32_t compute_table_size(int32_t input_size)
{
    uint32_t x = (uint32_t)(input_size - 1);
    int32_t target = 0;
    if (x != 0) {
        uint32_t hb = 0x80000000u >> __builtin_clz(x);
        target = (int32_t)(hb << 1);
    }
    if (target < MIN_TABLE_SIZE) return MIN_TABLE_SIZE;
    if (target > MAX_TABLE_SIZE) return MAX_TABLE_SIZE;
    return target;
}

/* Lz4RawCompressor.encodeRunLength :282-302 */
static int64_t encode_run_length(uint8_t* out, int64_t o, int64_t length)
{
    if (length >= RUN_MASK) {
        out[o++] = (uint8_t)(RUN_MASK << ML_BITS);
        int64_t remaining = length - RUN_MASK;
        while (remaining >= 255) {
            out[o++] = 255;
            remaining -= 255;
        }
        out[o++] = (uint8_t)remaining;
    }
    else {
        out[o++] = (uint8_t)(length << ML_BITS);
    }
    return o;
}

/* Lz4RawCompressor.emitLastLiteral :269-280 */
static int64_t emit_last_literal(uint8_t* out, int64_t o, const uint8_t* in, int64_t from, int64_t length)
{
    o = encode_run_length(out, o, length);
    if (length > 0) {  /* (an empty input has a null `in`) */
        memcpy(out + o, in + from, (size_t)length);
    }
    return o + length;
}

/* Lz4RawCompressor.emitLiteral :194-207 -- the 8-byte over-copy only touches bytes the
 * following emitMatch/emitLiteral own, so an exact copy yields the same final stream. */
static int64_t emit_literal(uint8_t* out, int64_t token_pos, const uint8_t* in, int64_t from, int32_t literal_length)
{
    int64_t o = encode_run_length(out, token_pos, literal_length);
    memcpy(out + o, in + from, (size_t)literal_length);
    return o + literal_length;
}

/* Lz4RawCompressor.emitMatch :209-235 */
static int64_t emit_match(uint8_t* out, int64_t o, int64_t token_pos, uint16_t offset, int64_t match_length)
{
    out[o] = (uint8_t)offset;
    out[o + 1] = (uint8_t)(offset >> 8);
    o += 2;
    if (match_length >= ML_MASK) {
        out[token_pos] |= ML_MASK;
        int64_t remaining = match_length - ML_MASK;
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
    else {
        out[token_pos] |= (uint8_t)match_length;
    }
    return o;
}

/* Lz4RawCompressor.count :240-267 */
static int32_t lz4_count(const uint8_t* in, int64_t input, int64_t limit, int64_t match)
{
    int32_t remaining = (int32_t)(limit - input);
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
        match++;
        input++;
    }
    return count;
}

/* Lz4RawCompressor.compress :69-192 */
int64_t orc_lz4_compress(const uint8_t* in, int64_t in_len64, uint8_t* out, int64_t out_cap)
{
    int32_t table[MAX_TABLE_SIZE];
    if (in_len64 > MAX_INPUT_SIZE) {
        return ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_LZ4_MAX_INPUT);
    }
    int32_t in_len = (int32_t)in_len64;
    int32_t table_size = compute_table_size(in_len);
    memset(table, 0, sizeof(int32_t) * (size_t)table_size);
    int32_t mask = table_size - 1;

    if (out_cap < orc_lz4_max_compressed_length(in_len)) {
        return ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_MAX_OUTPUT);
    }

    int64_t input = 0;
    int64_t output = 0;
    const int64_t input_limit = in_len;
    const int64_t match_find_limit = input_limit - MATCH_FIND_LIMIT;
    const int64_t match_limit = input_limit - LAST_LITERAL_SIZE;

    if (in_len < MIN_LENGTH) {
        return emit_last_literal(out, output, in, input, input_limit - input);
    }

    int64_t anchor = input;
    table[lz4_hash(ld64(in + input), mask)] = (int32_t)input;
    input++;
    int32_t next_hash = lz4_hash(ld64(in + input), mask);

    int done = 0;
    do {
        int64_t next_input_index = input;
        int32_t find_match_attempts = 1 << SKIP_TRIGGER;
        int32_t step = 1;
        int64_t match_index;
        do {
            int32_t hash = next_hash;
            input = next_input_index;
            next_input_index += step;
            step = (int32_t)((uint32_t)(find_match_attempts++) >> SKIP_TRIGGER);
            if (next_input_index > match_find_limit) {
                return emit_last_literal(out, output, in, anchor, input_limit - anchor);
            }
            match_index = table[hash];
            next_hash = lz4_hash(ld64(in + next_input_index), mask);
            table[hash] = (int32_t)input;
        }
        while (ld32(in + match_index) != ld32(in + input) || match_index + MAX_DISTANCE < input);

        /* catch up */
        while (input > anchor && match_index > 0 && in[input - 1] == in[match_index - 1]) {
            --input;
            --match_index;
        }

        int32_t literal_length = (int32_t)(input - anchor);
        int64_t token_pos = output;
        output = emit_literal(out, token_pos, in, anchor, literal_length);

        for (;;) {
            int32_t match_length = lz4_count(in, input + MIN_MATCH, match_limit, match_index + MIN_MATCH);
            output = emit_match(out, output, token_pos, (uint16_t)(input - match_index), match_length);
            input += match_length + MIN_MATCH;
            anchor = input;
            if (input > match_find_limit) {
                done = 1;
                break;
            }
            int64_t position = input - 2;
            table[lz4_hash(ld64(in + position), mask)] = (int32_t)position;

            int32_t hash = lz4_hash(ld64(in + input), mask);
            match_index = table[hash];
            table[hash] = (int32_t)input;
            if (match_index + MAX_DISTANCE < input || ld32(in + match_index) != ld32(in + input)) {
                input++;
                next_hash = lz4_hash(ld64(in + input), mask);
                break;
            }
            token_pos = output++;
            out[token_pos] = 0;
        }
    }
    while (!done);

    return emit_last_literal(out, output, in, anchor, input_limit - anchor);
}

/* Lz4RawDecompressor.decompress :35-198.  Copies are byte-exact LZ77 semantics: the
 * reference's 8-byte wild copies and DEC tables (:27-28,146-194) only ever disturb
 * bytes that a later write of the same call owns (or bytes past the returned length). */
int64_t orc_lz4_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off)
{
#define FAIL(detail, off)                                      \
    do {                                                       \
        if (err_off) *err_off = (off);                         \
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, (detail)); \
    } while (0)

    const int64_t input_limit = in_len;
    const int64_t output_limit = out_cap;
    const int64_t fast_output_limit = output_limit - 8;
    int64_t input = 0;
    int64_t output = 0;

    if (in_len == 0) {
        FAIL(ACHIP_D_LZ4_INPUT_EMPTY, 0);
    }
    if (out_cap == 0) {
        if (in_len == 1 && in[0] == 0) {
            return 0;
        }
        if (err_off) *err_off = 0;
        return ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT); /* Java returns -1 */
    }

    while (input < input_limit) {
        const int32_t token = in[input++];

        int32_t literal_length = token >> 4;
        if (literal_length == 0xF) {
            if (input >= input_limit) {
                FAIL(ACHIP_D_LZ4_MALFORMED, input);
            }
            int32_t value;
            do {
                value = in[input++];
                literal_length = (int32_t)((uint32_t)literal_length + (uint32_t)value);
            }
            while (value == 255 && input < input_limit - 15);
        }
        if (literal_length < 0) {
            FAIL(ACHIP_D_LZ4_MALFORMED, input);
        }

        int64_t literal_end = input + literal_length;
        int64_t literal_output_limit = output + literal_length;
        if (literal_output_limit > (fast_output_limit - MIN_MATCH) || literal_end > input_limit - (2 + 1 + LAST_LITERAL_SIZE)) {
            if (literal_output_limit > output_limit) {
                FAIL(ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE, input);
            }
            if (literal_end != input_limit) {
                FAIL(ACHIP_D_LZ4_INPUT_NOT_CONSUMED, input);
            }
            memcpy(out + output, in + input, (size_t)literal_length);
            output += literal_length;
            break;
        }

        memcpy(out + output, in + input, (size_t)literal_length);
        output = literal_output_limit;
        input = literal_end;

        int32_t offset = ld16(in + input);
        input += 2;

        int64_t match_address = output - offset;
        if (match_address < 0 || match_address >= output) {
            FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, input);
        }

        int32_t match_length = token & 0xF;
        if (match_length == 0xF) {
            int32_t value;
            do {
                if (input > input_limit - LAST_LITERAL_SIZE) {
                    FAIL(ACHIP_D_LZ4_MALFORMED, input);
                }
                value = in[input++];
                match_length = (int32_t)((uint32_t)match_length + (uint32_t)value);
            }
            while (value == 255);
        }
        match_length = (int32_t)((uint32_t)match_length + MIN_MATCH);
        if (match_length < 0) {
            FAIL(ACHIP_D_LZ4_MALFORMED, input);
        }

        int64_t match_output_limit = output + match_length;
        if (match_output_limit > fast_output_limit - MIN_MATCH) {
            if (match_output_limit > output_limit - LAST_LITERAL_SIZE) {
                FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, input);
            }
        }
        for (int64_t i = 0; i < match_length; i++) {
            out[output + i] = out[match_address + i];
        }
        output = match_output_limit;
    }
    return output;
#undef FAIL
}
