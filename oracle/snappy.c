/*
 * oracle/snappy.c -- CPU restatement of the reference's Java Snappy raw codec.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   compress   follows M/snappy/SnappyRawCompressor.java:74-411
 *   decompress follows M/snappy/SnappyRawDecompressor.java:35-322
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <string.h>

static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint16_t ld16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

enum {
    BLOCK_LOG = 16,
    BLOCK_SIZE = 1 << BLOCK_LOG,
    INPUT_MARGIN_BYTES = 15,
    MAX_HASH_TABLE_BITS = 14,
    MAX_HASH_TABLE_SIZE = 1 << MAX_HASH_TABLE_BITS,
    LITERAL = 0,
    COPY_1_BYTE_OFFSET = 1,
    COPY_2_BYTE_OFFSET = 2
};

/* SnappyRawCompressor.maxCompressedLength :47-70 */
int64_t orc_snappy_max_compressed_length(int64_t n) { return 32 + n + n / 6; }

/* SnappyRawCompressor.getHashTableSize :348-361 */
static int32_t get_hash_table_size(int32_t input_size)
{
    uint32_t x = (uint32_t)(input_size - 1);
    int32_t target = 0;
    if (x != 0) {
        uint32_t hb = 0x80000000u >> __builtin_clz(x);
        target = (int32_t)(hb << 1);
    }
    if (target < 256) return 256;
    if (target > MAX_HASH_TABLE_SIZE) return MAX_HASH_TABLE_SIZE;
    return target;
}

/* SnappyRawCompressor.hashBytes :368-371 */
static inline int32_t hash_bytes(uint32_t value, int32_t shift) { return (int32_t)((value * 0x1e35a7bdu) >> shift); }

/* SnappyRawCompressor.writeUncompressedLength :383-411 */
static int64_t write_uncompressed_length(uint8_t* out, int64_t o, int32_t n)
{
    const int32_t HB = 0x80;
    if (n < (1 << 7) && n >= 0) {
        out[o++] = (uint8_t)n;
    }
    else if (n < (1 << 14) && n > 0) {
        out[o++] = (uint8_t)(n | HB);
        out[o++] = (uint8_t)((uint32_t)n >> 7);
    }
    else if (n < (1 << 21) && n > 0) {
        out[o++] = (uint8_t)(n | HB);
        out[o++] = (uint8_t)(((uint32_t)n >> 7) | HB);
        out[o++] = (uint8_t)((uint32_t)n >> 14);
    }
    else if (n < (1 << 28) && n > 0) {
        out[o++] = (uint8_t)(n | HB);
        out[o++] = (uint8_t)(((uint32_t)n >> 7) | HB);
        out[o++] = (uint8_t)(((uint32_t)n >> 14) | HB);
        out[o++] = (uint8_t)((uint32_t)n >> 21);
    }
    else {
        out[o++] = (uint8_t)(n | HB);
        out[o++] = (uint8_t)(((uint32_t)n >> 7) | HB);
        out[o++] = (uint8_t)(((uint32_t)n >> 14) | HB);
        out[o++] = (uint8_t)(((uint32_t)n >> 21) | HB);
        out[o++] = (uint8_t)((uint32_t)n >> 28);
    }
    return o;
}

/* SnappyRawCompressor.count :235-266 (= exact common-prefix length capped at match_limit) */
static int32_t snappy_count(const uint8_t* in, int64_t start, int64_t match_start, int64_t match_limit)
{
    int64_t current = start;
    while (current < match_limit - 7) {
        uint64_t diff = ld64(in + match_start) ^ ld64(in + current);
        if (diff != 0) {
            current += __builtin_ctzll(diff) >> 3;
            return (int32_t)(current - start);
        }
        current += 8;
        match_start += 8;
    }
    if (current < match_limit - 3 && ld32(in + match_start) == ld32(in + current)) {
        current += 4;
        match_start += 4;
    }
    if (current < match_limit - 1 && ld16(in + match_start) == ld16(in + current)) {
        current += 2;
        match_start += 2;
    }
    if (current < match_limit && in[match_start] == in[current]) {
        ++current;
    }
    return (int32_t)(current - start);
}

/* SnappyRawCompressor.emitLiteralLength :268-298 -- the 4-byte store of n followed by
 * advancing `bytes` leaves exactly the low `bytes` bytes of n. */
static int64_t emit_literal_length(uint8_t* out, int64_t o, int32_t literal_length)
{
    int32_t n = literal_length - 1;
    if (n < 60) {
        out[o++] = (uint8_t)(n << 2);
    }
    else {
        int32_t bytes;
        if (n < (1 << 8)) {
            out[o++] = (uint8_t)((59 + 1) << 2);
            bytes = 1;
        }
        else if (n < (1 << 16)) {
            out[o++] = (uint8_t)((59 + 2) << 2);
            bytes = 2;
        }
        else if (n < (1 << 24)) {
            out[o++] = (uint8_t)((59 + 3) << 2);
            bytes = 3;
        }
        else {
            out[o++] = (uint8_t)((59 + 4) << 2);
            bytes = 4;
        }
        for (int32_t i = 0; i < bytes; i++) {
            out[o + i] = (uint8_t)((uint32_t)n >> (8 * i));
        }
        o += bytes;
    }
    return o;
}

/* SnappyRawCompressor.emitCopy :312-345 */
static int64_t emit_copy(uint8_t* out, int64_t o, int64_t input, int64_t match_index, int32_t match_length)
{
    int64_t offset = input - match_index;
    while (match_length >= 68) {
        out[o++] = (uint8_t)(COPY_2_BYTE_OFFSET + ((64 - 1) << 2));
        out[o++] = (uint8_t)offset;
        out[o++] = (uint8_t)(offset >> 8);
        match_length -= 64;
    }
    if (match_length > 64) {
        out[o++] = (uint8_t)(COPY_2_BYTE_OFFSET + ((60 - 1) << 2));
        out[o++] = (uint8_t)offset;
        out[o++] = (uint8_t)(offset >> 8);
        match_length -= 60;
    }
    if (match_length < 12 && offset < 2048) {
        int32_t len_minus4 = match_length - 4;
        out[o++] = (uint8_t)(COPY_1_BYTE_OFFSET + (len_minus4 << 2) + ((offset >> 8) << 5));
        out[o++] = (uint8_t)offset;
    }
    else {
        out[o++] = (uint8_t)(COPY_2_BYTE_OFFSET + ((match_length - 1) << 2));
        out[o++] = (uint8_t)offset;
        out[o++] = (uint8_t)(offset >> 8);
    }
    return o;
}

/* SnappyRawCompressor.compress :74-233 */
int64_t orc_snappy_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap)
{
    uint16_t table[MAX_HASH_TABLE_SIZE];
    int64_t max_len = orc_snappy_max_compressed_length((int32_t)in_len);
    if (out_cap < max_len) {
        return ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNAPPY_MAX_OUTPUT);
    }
    const int64_t input_limit = in_len;
    int64_t output = write_uncompressed_length(out, 0, (int32_t)in_len);

    for (int64_t block_address = 0; block_address < input_limit; block_address += BLOCK_SIZE) {
        const int64_t block_limit = (input_limit < block_address + BLOCK_SIZE) ? input_limit : block_address + BLOCK_SIZE;
        int64_t input = block_address;

        int32_t table_size = get_hash_table_size((int32_t)(block_limit - block_address));
        memset(table, 0, sizeof(uint16_t) * (size_t)table_size);
        const int32_t shift = 32 - (31 - __builtin_clz((uint32_t)table_size));

        int64_t next_emit = input;
        const int64_t fast_input_limit = block_limit - INPUT_MARGIN_BYTES;
        while (input <= fast_input_limit) {
            int32_t skip = 32;
            int64_t candidate = 0;
            for (input += 1; input + ((uint32_t)skip >> 5) <= fast_input_limit; input += ((uint32_t)(skip++) >> 5)) {
                uint32_t current_int = ld32(in + input);
                int32_t hash = hash_bytes(current_int, shift);
                candidate = block_address + table[hash];
                table[hash] = (uint16_t)(input - block_address);
                if (current_int == ld32(in + candidate)) {
                    break;
                }
            }
            if (input + ((uint32_t)skip >> 5) > fast_input_limit) {
                break;
            }

            int32_t literal_length = (int32_t)(input - next_emit);
            output = emit_literal_length(out, output, literal_length);
            memcpy(out + output, in + next_emit, (size_t)literal_length); /* fastCopy :300-310, exact bytes */
            output += literal_length;

            uint32_t input_bytes;
            do {
                int32_t matched = snappy_count(in, input + 4, candidate + 4, block_limit);
                matched += 4;
                output = emit_copy(out, output, input, candidate, matched);
                input += matched;
                if (input >= fast_input_limit) {
                    break;
                }
                uint64_t long_value = ld64(in + input - 1);
                uint32_t prev_int = (uint32_t)long_value;
                input_bytes = (uint32_t)(long_value >> 8);

                int32_t prev_hash = hash_bytes(prev_int, shift);
                table[prev_hash] = (uint16_t)(input - block_address - 1);

                int32_t cur_hash = hash_bytes(input_bytes, shift);
                candidate = block_address + table[cur_hash];
                table[cur_hash] = (uint16_t)(input - block_address);
            }
            while (input_bytes == ld32(in + candidate));
            next_emit = input;
        }

        if (next_emit < block_limit) {
            int32_t literal_length = (int32_t)(block_limit - next_emit);
            output = emit_literal_length(out, output, literal_length);
            memcpy(out + output, in + next_emit, (size_t)literal_length);
            output += literal_length;
        }
    }
    return output;
}

/* SnappyRawDecompressor.opLookupTable :227-271, regenerated from its documented layout:
 * bits 0-7 length, bits 8-10 copy-offset/256, bits 11-13 trailer byte count. */
static uint16_t op_lookup(int32_t op)
{
    int32_t kind = op & 3;
    int32_t hi = op >> 2;
    if (kind == LITERAL) {
        if (hi < 60) return (uint16_t)(hi + 1);
        return (uint16_t)(((hi - 59) << 11) | 1);
    }
    if (kind == COPY_1_BYTE_OFFSET) {
        return (uint16_t)((1 << 11) | ((hi >> 3) << 8) | ((hi & 7) + 4));
    }
    if (kind == COPY_2_BYTE_OFFSET) {
        return (uint16_t)((2 << 11) | (hi + 1));
    }
    return (uint16_t)((4 << 11) | (hi + 1));
}

/* SnappyRawDecompressor.readUncompressedLength :277-321.  Returns length (>= 0) and
 * *bytes_read, or a status with *err_off. Java's "offset" here is an absolute address
 * expression (compressedAddress + bytesRead / limit - address); we report it relative
 * to the input start, which is what those expressions evaluate to for address 0. */
static int64_t read_uncompressed_length(const uint8_t* in, int64_t in_len, int32_t* bytes_read, int64_t* err_off)
{
    uint32_t result = 0;
    int32_t n = 0;
    for (int32_t i = 0; i < 5; i++) {
        if (n >= in_len) {
            if (err_off) *err_off = in_len - n; /* limit - address */
            return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_TRUNCATED);
        }
        uint32_t b = in[n++];
        result |= (b & 0x7f) << (7 * i);
        if ((b & 0x80) == 0) {
            break;
        }
        if (i == 4) {
            if (err_off) *err_off = n;
            return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LEN_HIGH_BIT);
        }
    }
    if ((int32_t)result < 0) {
        if (err_off) *err_off = 0;
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_INVALID_LENGTH);
    }
    *bytes_read = n;
    return (int64_t)result;
}

int64_t orc_snappy_uncompressed_length(const uint8_t* in, int64_t in_len, int64_t* err_off)
{
    int32_t n;
    return read_uncompressed_length(in, in_len, &n, err_off);
}

/* SnappyRawDecompressor.decompress :35-68 + uncompressAll :70-220 */
int64_t orc_snappy_decompress(const uint8_t* in0, int64_t in_len0, uint8_t* out, int64_t out_cap, int64_t* err_off)
{
#define FAIL(off)                                                             \
    do {                                                                      \
        if (err_off) *err_off = (off);                                        \
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_MALFORMED); \
    } while (0)

    int32_t varint_bytes = 0;
    int64_t expected = read_uncompressed_length(in0, in_len0, &varint_bytes, err_off);
    if (expected < 0) {
        return expected;
    }
    if (expected > out_cap) {
        if (err_off) *err_off = 0;
        return ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNAPPY_OUTPUT_TOO_SMALL);
    }

    /* uncompressAll: offsets are relative to the first byte after the varint (:70-76) */
    const uint8_t* in = in0 + varint_bytes;
    const int64_t input_limit = in_len0 - varint_bytes;
    const int64_t output_limit = out_cap;
    const int64_t fast_output_limit = output_limit - 8;
    static const uint32_t wordmask[5] = {0, 0xff, 0xffff, 0xffffff, 0xffffffffu};

    int64_t output = 0;
    int64_t input = 0;
    while (input < input_limit) {
        int32_t op = in[input++];
        int32_t entry = op_lookup(op);
        int32_t trailer_bytes = entry >> 11;
        int32_t trailer = 0;
        if (input + 4 < input_limit) {
            trailer = (int32_t)(ld32(in + input) & wordmask[trailer_bytes]);
        }
        else {
            if (input + trailer_bytes > input_limit) {
                FAIL(input);
            }
            uint32_t t = 0;
            switch (trailer_bytes) {
                case 4: t = (uint32_t)in[input + 3] << 24; /* fallthrough */
                case 3: t |= (uint32_t)in[input + 2] << 16; /* fallthrough */
                case 2: t |= (uint32_t)in[input + 1] << 8;  /* fallthrough */
                case 1: t |= (uint32_t)in[input];
                default: break;
            }
            trailer = (int32_t)t;
        }
        if (trailer < 0) {
            FAIL(input);
        }
        input += trailer_bytes;

        int32_t length = entry & 0xff;
        if (length == 0) {
            continue;
        }

        if ((op & 3) == LITERAL) {
            int32_t literal_length = (int32_t)((uint32_t)length + (uint32_t)trailer);
            if (literal_length < 0) {
                FAIL(input);
            }
            int64_t literal_output_limit = output + literal_length;
            if (literal_output_limit > fast_output_limit || input + literal_length > input_limit - 8) {
                if (literal_output_limit > output_limit || input + literal_length > input_limit) {
                    FAIL(input);
                }
            }
            memcpy(out + output, in + input, (size_t)literal_length);
            input += literal_length;
            output += literal_length;
        }
        else {
            int32_t match_offset = entry & 0x700;
            match_offset = (int32_t)((uint32_t)match_offset + (uint32_t)trailer);
            if (match_offset <= 0) {
                FAIL(input);
            }
            int64_t match_address = output - match_offset;
            if (match_address < 0 || output + length > output_limit) {
                FAIL(input);
            }
            for (int32_t i = 0; i < length; i++) {
                out[output + i] = out[match_address + i];
            }
            output += length;
        }
    }

    if (expected != output) {
        if (err_off) *err_off = 0;
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LENGTH_MISMATCH);
    }
    return expected;
#undef FAIL
}
