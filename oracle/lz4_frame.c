/*
 * oracle/lz4_frame.c -- the LZ4 frame container as the reference implements it.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Follows M/lz4/Lz4FrameCompression.java: maxCompressedLength :70-83, compress :96-140, decompress :145-177,
 * decompressFrame :184-322, skipFrame :327-343; constants M/lz4/Lz4FrameFormat.java:24-68.  Blocks go through
 * orc_lz4_compress / orc_lz4_decompress (the Java block codec the *Java* frame classes delegate to).
 * Pinning: the reference's tests build their vectors in code (T/lz4/TestLz4FrameDecompressor.java:61-230); the CPU suite
 * rebuilds the same vectors (tests/test_oracle_lz4_frame.py) and cross-decodes both ways against liblz4's frame codec (pyarrow).
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <stdlib.h>
#include <string.h>

#define MAGIC 0x184D2204u
#define SKIPPABLE_MAGIC 0x184D2A50u
#define SKIPPABLE_MASK 0xFFFFFFF0u
#define FLG_VERSION (1 << 6)
#define FLG_BLOCK_INDEPENDENCE (1 << 5)
#define FLG_BLOCK_CHECKSUM (1 << 4)
#define FLG_CONTENT_SIZE (1 << 3)
#define FLG_CONTENT_CHECKSUM (1 << 2)
#define FLG_DICTIONARY_ID 1
#define FLG_RESERVED_MASK 0x02
#define BD_RESERVED_MASK 0x8F
#define BD_4MB (7 << 4)
#define BLOCK_MAX_4MB (4 * 1024 * 1024)
#define HEADER_SIZE 7
#define END_MARK_SIZE 4
#define UNCOMPRESSED_FLAG 0x80000000u
#define MALFORMED(d) ACHIP_STATUS(ACHIP_CLASS_MALFORMED, d)

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static void wr32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }

int64_t orc_lz4frame_max_compressed_length(int64_t n) /* :70-83 */
{
    if (n < 0) return ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
    const int64_t blocks = (n + BLOCK_MAX_4MB - 1) / BLOCK_MAX_4MB;
    const int64_t max = HEADER_SIZE + END_MARK_SIZE + n + 4 * blocks;
    if (max > 0x7FFFFFFF) return ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
    return max;
}

#define TOO_SMALL ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4F_MAX_OUTPUT)

int64_t orc_lz4frame_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap) /* :96-140 */
{
    int64_t pos = 0;
    if (pos + 4 > out_cap) return TOO_SMALL;
    wr32(out + pos, MAGIC);
    pos += 4;
    if (pos + 1 > out_cap) return TOO_SMALL;
    out[pos++] = (uint8_t)(FLG_VERSION | FLG_BLOCK_INDEPENDENCE);
    if (pos + 1 > out_cap) return TOO_SMALL;
    out[pos++] = (uint8_t)BD_4MB;
    const uint32_t hc = (orc_xxh32(out + pos - 2, 2, 0) >> 8) & 0xFF;
    if (pos + 1 > out_cap) return TOO_SMALL;
    out[pos++] = (uint8_t)hc;

    int64_t clamp = in_len < 1 ? 1 : (in_len > BLOCK_MAX_4MB ? BLOCK_MAX_4MB : in_len);
    const int64_t scratch_cap = orc_lz4_max_compressed_length(clamp);
    uint8_t* scratch = (uint8_t*)malloc((size_t)scratch_cap);
    int64_t ipos = 0;
    int64_t result = 0;
    while (ipos < in_len) {
        const int64_t block_len = in_len - ipos < BLOCK_MAX_4MB ? in_len - ipos : BLOCK_MAX_4MB;
        const int64_t clen = orc_lz4_compress(in + ipos, block_len, scratch, scratch_cap);
        if (clen < 0) { result = clen; goto done; }
        if (clen < block_len) {
            if (pos + 4 > out_cap) { result = TOO_SMALL; goto done; }
            wr32(out + pos, (uint32_t)clen);
            pos += 4;
            if (pos + clen > out_cap) { result = TOO_SMALL; goto done; }
            memcpy(out + pos, scratch, (size_t)clen);
            pos += clen;
        }
        else {
            if (pos + 4 > out_cap) { result = TOO_SMALL; goto done; }
            wr32(out + pos, (uint32_t)block_len | UNCOMPRESSED_FLAG);
            pos += 4;
            if (pos + block_len > out_cap) { result = TOO_SMALL; goto done; }
            memcpy(out + pos, in + ipos, (size_t)block_len);
            pos += block_len;
        }
        ipos += block_len;
    }
    if (pos + 4 > out_cap) { result = TOO_SMALL; goto done; }
    wr32(out + pos, 0);
    pos += 4;
    result = pos;
done:
    free(scratch);
    return result;
}

static int32_t block_maximum_size(int id) /* Lz4FrameFormat.java:58-67 */
{
    switch (id) {
        case 4: return 64 * 1024;
        case 5: return 256 * 1024;
        case 6: return 1024 * 1024;
        case 7: return 4 * 1024 * 1024;
        default: return -1;
    }
}

#define FAIL(d, off)          \
    {                         \
        *err_off = (off);     \
        return MALFORMED(d);  \
    }

/* decompressFrame :184-322 ; returns 0 or status; *ipos / *opos advance */
static int64_t decompress_frame(const uint8_t* in, int64_t in_len, int64_t frame_start, uint8_t* out, int64_t out_cap, int64_t out_start, int64_t* ipos, int64_t* opos,
                                int64_t* err_off)
{
    const int64_t dstart = frame_start + 4;
    if (dstart + 2 > in_len) FAIL(ACHIP_D_LZ4F_TRUNC_HEADER, dstart);
    const int flg = in[dstart], bd = in[dstart + 1];
    const int version = (flg >> 6) & 3;
    if (version != 1) FAIL(version == 0 ? ACHIP_D_LZ4F_VERSION_0 : (version == 2 ? ACHIP_D_LZ4F_VERSION_2 : ACHIP_D_LZ4F_VERSION_3), dstart);
    if ((flg & FLG_RESERVED_MASK) != 0 || (bd & BD_RESERVED_MASK) != 0) FAIL(ACHIP_D_LZ4F_RESERVED_BITS, dstart);
    const int block_checksum = (flg & FLG_BLOCK_CHECKSUM) != 0, content_size = (flg & FLG_CONTENT_SIZE) != 0, content_checksum = (flg & FLG_CONTENT_CHECKSUM) != 0;
    if ((flg & FLG_BLOCK_INDEPENDENCE) == 0) FAIL(ACHIP_D_LZ4F_LINKED_BLOCKS, dstart);
    if ((flg & FLG_DICTIONARY_ID) != 0) FAIL(ACHIP_D_LZ4F_DICTIONARY, dstart);
    const int32_t bmax = block_maximum_size((bd >> 4) & 7);
    if (bmax < 0) FAIL(ACHIP_D_LZ4F_BLOCK_MAX_SIZE, dstart + 1);
    int64_t pos = dstart + 2;
    if (pos + (content_size ? 8 : 0) + 1 > in_len) FAIL(ACHIP_D_LZ4F_TRUNC_HEADER, pos);
    int64_t expected_size = -1;
    if (content_size) {
        memcpy(&expected_size, in + pos, 8);
        pos += 8;
    }
    const int expected_hc = in[pos];
    const int actual_hc = (int)((orc_xxh32(in + dstart, pos - dstart, 0) >> 8) & 0xFF);
    if (expected_hc != actual_hc) FAIL(ACHIP_D_LZ4F_HEADER_CHECKSUM, pos);
    pos++;

    int64_t op = out_start;
    for (;;) {
        if (pos + 4 > in_len) FAIL(ACHIP_D_LZ4F_MISSING_BLOCK_SIZE, pos);
        const uint32_t header = rd32(in + pos);
        pos += 4;
        if (header == 0) break;
        const int uncompressed = (header & UNCOMPRESSED_FLAG) != 0;
        const int64_t block_len = header & 0x7FFFFFFFu;
        if (block_len > bmax || pos + block_len > in_len) FAIL(ACHIP_D_LZ4F_BLOCK_PAST_END, pos);
        if (uncompressed) {
            if (op + block_len > out_cap) FAIL(ACHIP_D_LZ4F_OUTPUT_TOO_SMALL, op);
            memcpy(out + op, in + pos, (size_t)block_len);
            op += block_len;
        }
        else {
            int64_t beo = 0;
            const int64_t n = orc_lz4_decompress(in + pos, block_len, out + op, out_cap - op, &beo);
            if (n < 0) {
                if (-n == ACHIP_CLASS_OUTPUT_TOO_SMALL + 16 * ACHIP_D_LZ4_EMPTY_OUTPUT) FAIL(ACHIP_D_LZ4F_OUTPUT_TOO_SMALL, op); /* the block codec returned -1 :275-277 */
                *err_off = beo; /* the block codec's exception propagates, offset relative to the block */
                return n;
            }
            if (n > bmax) FAIL(ACHIP_D_LZ4F_BLOCK_EXCEEDS_MAX, pos);
            op += n;
        }
        if (block_checksum) {
            const int64_t cpos = pos + block_len;
            if (cpos + 4 > in_len) FAIL(ACHIP_D_LZ4F_MISSING_BLOCK_CHECKSUM, cpos);
            if (rd32(in + cpos) != orc_xxh32(in + pos, block_len, 0)) FAIL(ACHIP_D_LZ4F_BLOCK_CHECKSUM, cpos);
        }
        pos += block_len;
        if (block_checksum) pos += 4;
    }
    const int64_t content_len = op - out_start;
    if (content_checksum) {
        if (pos + 4 > in_len) FAIL(ACHIP_D_LZ4F_MISSING_CONTENT_CHECKSUM, pos);
        if (rd32(in + pos) != orc_xxh32(out + out_start, content_len, 0)) FAIL(ACHIP_D_LZ4F_CONTENT_CHECKSUM, pos);
        pos += 4;
    }
    if (content_size && content_len != expected_size) FAIL(ACHIP_D_LZ4F_CONTENT_SIZE, pos);
    *ipos = pos;
    *opos = op;
    return 0;
}

int64_t orc_lz4frame_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off) /* :145-177 */
{
    *err_off = 0;
    if (in_len < HEADER_SIZE) FAIL(ACHIP_D_LZ4F_TOO_SHORT, 0);
    int64_t pos = 0, op = 0;
    while (pos < in_len) {
        if (pos + 4 > in_len) FAIL(ACHIP_D_LZ4F_TRUNC_MAGIC, pos);
        const uint32_t magic = rd32(in + pos);
        if (magic == MAGIC) {
            const int64_t r = decompress_frame(in, in_len, pos, out, out_cap, op, &pos, &op, err_off);
            if (r < 0) return r;
        }
        else if ((magic & SKIPPABLE_MASK) == SKIPPABLE_MAGIC) { /* skipFrame :327-343 */
            const int64_t spos = pos + 4;
            if (spos + 4 > in_len) FAIL(ACHIP_D_LZ4F_TRUNC_SKIP_SIZE, spos);
            const int64_t frame_end = spos + 4 + (int64_t)rd32(in + spos);
            if (frame_end > in_len) FAIL(ACHIP_D_LZ4F_TRUNC_SKIP, spos);
            pos = frame_end;
        }
        else {
            FAIL(ACHIP_D_LZ4F_BAD_MAGIC, pos);
        }
    }
    return op;
}
