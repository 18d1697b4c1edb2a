/*
 * oracle/snappy_framed.c -- x-snappy-framed streams as the reference's stream classes read and write them.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Whole-buffer restatement of
 *   M/snappy/SnappyFramedOutputStream.java: constructor :73-96 (stream header), write(byte[],int,int) :113-145, close :158-171,
 *       writeCompressed :200-222 (masked CRC-32C of the plaintext, block compressed, kept when compressed/length <= 0.85),
 *       writeBlock :236-255;  defaults: block size 65536, checksums on                                   (:33-37)
 *   M/snappy/SnappyFramedInputStream.java: constructor :52-73 (stream header), ensureBuffer :135-214, getFrameMetaData :226-277,
 *       getFrameData :279-288, readBlockHeader :290-305;  verifyChecksums = true
 *   M/snappy/SnappyFramed.java:24-31 (chunk types, header bytes), M/snappy/Crc32C.java:29-50 (masked CRC-32C; the Java tables are
 *       the slicing-by-8 tables of polynomial 0x1EDC6F41 reflected -- plain CRC-32C)
 *   M/snappy/SnappyInternalUtils.java:91-142 (readBytes: -1 at end of stream; skip: stops quietly at end of stream)
 * over orc_snappy_compress / orc_snappy_decompress (the Java block codec the stream classes are given in the tests).
 *
 * "compress" = new SnappyFramedOutputStream(c, out); write(all); close().  "decompress" = read the stream to its end.
 * Two things the one-shot form has to add (they have no Java counterpart): the destination capacity (ACHIP_D_SNF_OUTPUT_TOO_SMALL /
 * ACHIP_D_SNF_MAX_OUTPUT), and the offset reported with the stream-level IOExceptions, which carry none: the position of the chunk
 * header (of the stream header: 0).  A block codec exception keeps its own offset (relative to the chunk's data).
 * The Java reader decodes a chunk into a buffer of max(65541, everything seen so far) bytes; the same capacity is handed to the
 * block decoder here, limited by what the destination has left.  Java's getUncompressedLength may read past the chunk into stale
 * bytes of its buffer when the chunk ends inside the preamble; here the preamble ends with the chunk (ACHIP_D_SNAPPY_TRUNCATED).
 *
 * Pinning: T/snappy/TestSnappyStream.java:50-176 (sizes, flags, lengths and the CRC of testSimple; every error case) rebuilt in
 * tests/test_oracle_snappy_framed.py; CRC-32C against the RFC 3720 B.4 vectors.
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <stdlib.h>
#include <string.h>

#define COMPRESSED_DATA_FLAG 0x00
#define UNCOMPRESSED_DATA_FLAG 0x01
#define STREAM_IDENTIFIER_FLAG 0xff
#define MAX_BLOCK_SIZE 65536
#define MIN_COMPRESSION_RATIO 0.85
#define MALFORMED(d) ACHIP_STATUS(ACHIP_CLASS_MALFORMED, d)

static const uint8_t HEADER_BYTES[10] = {0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59};

/* CRC-32C (Castagnoli), reflected, init / final xor 0xFFFFFFFF -- what Crc32C.update + getValue compute */
uint32_t orc_crc32c(const uint8_t* p, int64_t n)
{
    static uint32_t table[256];
    static int ready = 0;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) {
                c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            }
            table[i] = c;
        }
        ready = 1;
    }
    uint32_t crc = 0xFFFFFFFFu;
    for (int64_t i = 0; i < n; i++) {
        crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    }
    return ~crc;
}

uint32_t orc_masked_crc32c(const uint8_t* p, int64_t n) /* Crc32C.java:34-50: rotate right by 15, add 0xa282ead8 */
{
    const uint32_t crc = orc_crc32c(p, n);
    return ((crc >> 15) | (crc << 17)) + 0xa282ead8u;
}

int64_t orc_snappyframed_max_compressed_length(int64_t n)
{
    if (n < 0) return ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
    /* a chunk is never larger than its block stored raw (compressed is kept only at <= 0.85 of it) */
    const int64_t blocks = (n + MAX_BLOCK_SIZE - 1) / MAX_BLOCK_SIZE;
    const int64_t max = 10 + 8 * blocks + n;
    return max > 0x7FFFFFFF ? ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT) : max;
}

int64_t orc_snappyframed_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap)
{
    const int64_t bound = orc_snappyframed_max_compressed_length(in_len);
    if (bound < 0) return bound;
    if (out_cap < bound) return ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_MAX_OUTPUT);
    memcpy(out, HEADER_BYTES, 10); /* constructor :94 */
    int64_t o = 10;
    const int64_t scratch_cap = orc_snappy_max_compressed_length(MAX_BLOCK_SIZE);
    uint8_t* scratch = (uint8_t*)malloc((size_t)scratch_cap);
    if (!scratch) return ACHIP_STATUS(ACHIP_CLASS_DEVICE, ACHIP_D_HIP_ERROR);
    /* write(all) + close: full blocks straight from the input, the rest through the buffer (:126-145, flushBuffer :186-192) */
    for (int64_t pos = 0; pos < in_len; pos += MAX_BLOCK_SIZE) {
        const int64_t length = in_len - pos < MAX_BLOCK_SIZE ? in_len - pos : MAX_BLOCK_SIZE;
        const uint32_t crc = orc_masked_crc32c(in + pos, length);                            /* :204 */
        const int64_t compressed = orc_snappy_compress(in + pos, length, scratch, scratch_cap); /* :206-211 */
        if (compressed < 0) {
            free(scratch);
            return compressed;
        }
        const int keep = ((double)compressed / (double)length) <= MIN_COMPRESSION_RATIO;      /* :214 */
        const uint8_t* data = keep ? scratch : in + pos;
        const int64_t dlen = keep ? compressed : length;
        const int64_t header_length = dlen + 4;                                              /* writeBlock :241-254 */
        out[o++] = keep ? COMPRESSED_DATA_FLAG : UNCOMPRESSED_DATA_FLAG;
        out[o++] = (uint8_t)header_length;
        out[o++] = (uint8_t)(header_length >> 8);
        out[o++] = (uint8_t)(header_length >> 16);
        out[o++] = (uint8_t)crc;
        out[o++] = (uint8_t)(crc >> 8);
        out[o++] = (uint8_t)(crc >> 16);
        out[o++] = (uint8_t)(crc >> 24);
        memcpy(out + o, data, (size_t)dlen);
        o += dlen;
    }
    free(scratch);
    return o;
}

int64_t orc_snappyframed_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off)
{
#define FAIL(detail, off)      \
    do {                       \
        *err_off = (off);      \
        return MALFORMED(detail); \
    } while (0)
    *err_off = 0;
    if (in_len < 10) FAIL(ACHIP_D_SNF_EOF_STREAM_HEADER, 0);             /* :66-69 */
    if (memcmp(in, HEADER_BYTES, 10) != 0) FAIL(ACHIP_D_SNF_BAD_STREAM_HEADER, 0); /* :70-72 */
    int64_t pos = 10;
    int64_t o = 0;
    int64_t java_input = MAX_BLOCK_SIZE + 5, java_uncompressed = MAX_BLOCK_SIZE + 5; /* allocateBuffersBasedOnSize(MAX_BLOCK_SIZE + 5) :61 */
    for (;;) {
        const int64_t chunk = pos;
        if (pos == in_len) break;                                         /* readBlockHeader: -1 => end of stream :295-297 */
        if (in_len - pos < 4) FAIL(ACHIP_D_SNF_EOF_BLOCK_HEADER, chunk);  /* :299-301 */
        const int flag = in[pos];
        const int64_t length = in[pos + 1] | (in[pos + 2] << 8) | ((int64_t)in[pos + 3] << 16);
        pos += 4;
        int skip = 0;
        int64_t min_length;
        if (flag == COMPRESSED_DATA_FLAG || flag == UNCOMPRESSED_DATA_FLAG) { /* getFrameMetaData :234-277 */
            min_length = 5;
        }
        else if (flag == STREAM_IDENTIFIER_FLAG) {
            if (length != 6) FAIL(ACHIP_D_SNF_STREAM_ID_LENGTH, chunk);
            skip = 1;
            min_length = 6;
        }
        else {
            if (flag <= 0x7f) FAIL(ACHIP_D_SNF_UNSKIPPABLE, chunk);
            skip = 1;
            min_length = 0;
        }
        if (length < min_length) FAIL(ACHIP_D_SNF_INVALID_LENGTH, chunk);
        if (skip) {                                                       /* :151-154; skip stops quietly at the end of the stream */
            pos += length < in_len - pos ? length : in_len - pos;
            continue;
        }
        if (length > java_input) {                                        /* :156-158 allocateBuffersBasedOnSize */
            java_input = length;
            if (java_uncompressed < length) java_uncompressed = length;
        }
        if (in_len - pos < length) FAIL(ACHIP_D_SNF_EOF_FRAME, chunk);    /* :160-163 */
        const uint8_t* content = in + pos;
        const uint32_t stored = content[0] | (content[1] << 8) | (content[2] << 16) | ((uint32_t)content[3] << 24); /* getFrameData */
        const uint8_t* data = content + 4;
        const int64_t dlen = length - 4;
        int64_t produced;
        if (flag == COMPRESSED_DATA_FLAG) {                               /* :167-177 */
            int64_t eo = 0;
            const int64_t ulen = orc_snappy_uncompressed_length(data, dlen, &eo);
            if (ulen < 0) {
                *err_off = eo;
                return ulen;
            }
            if (ulen > java_uncompressed) java_uncompressed = ulen;
            if (ulen > out_cap - o) {
                *err_off = chunk;
                return ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_OUTPUT_TOO_SMALL);
            }
            const int64_t limit = java_uncompressed < out_cap - o ? java_uncompressed : out_cap - o;
            produced = orc_snappy_decompress(data, dlen, out + o, limit, &eo);
            if (produced < 0) {
                *err_off = eo;
                return produced;
            }
        }
        else {                                                            /* raw :178-186 */
            if (dlen > out_cap - o) {
                *err_off = chunk;
                return ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_OUTPUT_TOO_SMALL);
            }
            memcpy(out + o, data, (size_t)dlen);
            produced = dlen;
        }
        if (stored != orc_masked_crc32c(out + o, produced)) FAIL(ACHIP_D_SNF_CHECKSUM, chunk); /* :188-193 */
        o += produced;
        pos += length;
    }
#undef FAIL
    return o;
}
