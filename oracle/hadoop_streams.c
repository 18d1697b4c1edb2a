/*
 * oracle/hadoop_streams.c -- Hadoop LZ4 / Snappy block streams as the reference's stream classes read and write them (SURVEY 8f row 2,
 * second half).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Whole-buffer restatement of
 *   M/lz4/Lz4HadoopOutputStream.java / M/snappy/SnappyHadoopOutputStream.java: write(byte[],int,int) :60-79, finish :82-88,
 *       writeNextChunk :107-118 ([BE int plaintext length][BE int compressed length][block]), compressionOverhead :128-131
 *       (LZ4: max((int)(size * 0.01), 10); Snappy: size / 6 + 32); buffer size from Lz4HadoopStreams.java:30 / SnappyHadoopStreams.java:30
 *       (256 KiB unless configured)
 *   M/lz4/Lz4HadoopInputStream.java: read() :47-58, read(byte[],int,int) :61-82, bufferCompressedData :100-127, readInput :129-140,
 *       readBigEndianInt :142-156
 *   M/snappy/SnappyHadoopInputStream.java: read() :44-54, read(byte[],int,int) :57-73, readNextChunk :91-141
 * driven the way the reference's own test harness drives them:
 *   compress   = T/HadoopCodecCompressor.java:57-72     createOutputStream; write(everything); close
 *   decompress = T/HadoopCodecDecompressor.java:40-60   read(output, done, capacity - done) until it returns -1 or the output is full, then
 *                one read(): a byte there is "All input was not consumed" (a RuntimeException in the harness;
 *                ACHIP_D_HDP_NOT_CONSUMED here -- the destination cannot hold the stream).
 * over orc_lz4_* / orc_snappy_* (the Java block codecs the streams are given).
 *
 * What the one-shot form adds (no Java counterpart): the offset reported with the stream-level IOExceptions, which carry none -- the
 * position in the stream where the read that failed began; a block codec exception keeps its own offset (relative to the chunk).
 * Deviations, both for inputs no writer produces and both in the Snappy reader only: a negative chunk length other than -1 is
 * ACHIP_D_HDP_NEGATIVE_LENGTH (SnappyHadoopInputStream.java:110-133 goes on to ask getUncompressedLength of whatever its buffer still holds from
 * the chunk before, then fails in the block codec's range check: an exception whose kind depends on stale state); and, because that
 * method is asked of the whole buffer, a chunk that ends inside its length preamble makes Java read stale bytes -- here the preamble
 * ends with the chunk (ACHIP_D_SNAPPY_TRUNCATED), as in snappy_framed.c.  (The LZ4 reader takes any negative chunk length for the end
 * of the stream, as Java does.)
 *
 * Pinning: the format has no golden vectors in the reference (its tests round-trip through org.apache.hadoop's codecs, absent here);
 * tests/test_oracle_hadoop.py checks the writer against a byte-level description of the format, the reader against hand-built streams
 * covering every branch above, and both against each other over the corpus.  Since round 3 the reference's OWN stream classes execute
 * here (oracle/_ref: transliterated by tools/j2c.py) and tests/test_ref_pin.py compares this file with them: the writers byte for byte,
 * the readers on every one of those hand-built streams and on random damage.
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <stdlib.h>
#include <string.h>

#define MALFORMED(d) ACHIP_STATUS(ACHIP_CLASS_MALFORMED, d)
#define STREAM_EOF (-1000000)  /* internal: end of stream (never returned to the caller) */

static int32_t input_max_size(int32_t codec, int32_t bufferSize)
{
    const int32_t overhead = codec == 0 ? (((int32_t)(bufferSize * 0.01)) > 10 ? (int32_t)(bufferSize * 0.01) : 10) : (bufferSize / 6) + 32;
    return bufferSize - overhead;
}

static int64_t block_max(int32_t codec, int64_t n) { return codec == 0 ? orc_lz4_max_compressed_length(n) : orc_snappy_max_compressed_length(n); }

int64_t orc_hadoop_max_compressed_length(int32_t codec, int64_t n, int32_t bufferSize)
{
    const int64_t chunk = input_max_size(codec, bufferSize);
    if (n < 0 || chunk <= 0) {
        return ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
    }
    const int64_t full = n / chunk, rest = n % chunk;
    return full * (8 + block_max(codec, chunk)) + (rest > 0 ? 8 + block_max(codec, rest) : 0);
}

static void put_be(uint8_t* p, int32_t v)
{
    p[0] = (uint8_t)((uint32_t)v >> 24);
    p[1] = (uint8_t)((uint32_t)v >> 16);
    p[2] = (uint8_t)((uint32_t)v >> 8);
    p[3] = (uint8_t)v;
}

/* new XHadoopOutputStream(compressor, out, bufferSize); write(in, 0, n); close() */
int64_t orc_hadoop_compress(int32_t codec, const uint8_t* in, int64_t n, uint8_t* out, int64_t cap, int32_t bufferSize)
{
    const int64_t bound = orc_hadoop_max_compressed_length(codec, n, bufferSize);
    if (bound < 0) {
        return bound;
    }
    if (cap < bound) {
        return ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_HDP_MAX_OUTPUT);
    }
    const int64_t chunk = input_max_size(codec, bufferSize);
    int64_t o = 0;
    for (int64_t pos = 0; pos < n; pos += chunk) {  /* write :60-79 -- whole chunks straight from the caller's buffer, the rest through finish :82-88 */
        const int64_t len = n - pos < chunk ? n - pos : chunk;
        const int64_t c = codec == 0 ? orc_lz4_compress(in + pos, len, out + o + 8, block_max(codec, len)) : orc_snappy_compress(in + pos, len, out + o + 8, block_max(codec, len));
        if (c < 0) {
            return c;
        }
        put_be(out + o, (int32_t)len);  /* writeNextChunk :107-118 */
        put_be(out + o + 4, (int32_t)c);
        o += 8 + c;
    }
    return o;
}

/* ---- the readers ---- */
typedef struct {
    int32_t codec, bufferSize;
    const uint8_t* in;
    int64_t n, pos;
    int64_t blockLen;            /* uncompressedBlockLength */
    int64_t chunkOff, chunkLen;  /* uncompressedChunkOffset / Length */
    uint8_t* internal;           /* uncompressedChunk */
    int64_t internalCap;
    const uint8_t* chunk;        /* `compressed` */
    int64_t eo;                  /* offset that goes with a failure */
} Reader;

/* readBigEndianInt :142-156 ; STREAM_EOF when the stream ends before the first byte (or the value is -1: the Java code cannot tell) */
static int64_t read_be(Reader* r, int32_t* v)
{
    if (r->pos >= r->n) {
        return STREAM_EOF;
    }
    if (r->pos + 4 > r->n) {
        r->eo = r->pos;
        return MALFORMED(ACHIP_D_HDP_TRUNCATED_INT);
    }
    const uint8_t* p = r->in + r->pos;
    *v = (int32_t)(((uint32_t)p[0] << 24) + ((uint32_t)p[1] << 16) + ((uint32_t)p[2] << 8) + (uint32_t)p[3]);
    r->pos += 4;
    return *v == -1 ? STREAM_EOF : 0;
}

/* the common head of bufferCompressedData :100-127 and readNextChunk :91-113 ; returns the chunk's compressed length, STREAM_EOF or a status */
static int64_t next_chunk(Reader* r)
{
    r->blockLen -= r->chunkOff;
    r->chunkOff = 0;
    r->chunkLen = 0;
    while (r->blockLen == 0) {
        int32_t v = 0;
        const int64_t e = read_be(r, &v);
        if (e == STREAM_EOF) {
            r->blockLen = 0;
            return STREAM_EOF;
        }
        if (e < 0) {
            return e;
        }
        r->blockLen = v;
    }
    int32_t clen = 0;
    const int64_t e = read_be(r, &clen);
    if (e != 0) {
        return e;
    }
    if (clen < 0) {
        if (r->codec == 0) {
            /* Lz4HadoopInputStream: bufferCompressedData :113-126 allocates nothing and reads nothing for a negative length and returns it; both
             * callers test `compressedChunkLength < 0` (:51-54, :65-68), so ANY negative value ends the stream for the read that sees it --
             * pinned by the reference's own classes executing (tests/test_ref_pin.py) */
            return STREAM_EOF;
        }
        r->eo = r->pos - 4;
        return MALFORMED(ACHIP_D_HDP_NEGATIVE_LENGTH);  /* Snappy: only -1 ends the stream (:105-108); see the header for the rest */
    }
    if (r->pos + clen > r->n) {  /* readInput :129-140 */
        r->eo = r->pos;
        return MALFORMED(ACHIP_D_HDP_EOF_BLOCK_DATA);
    }
    r->chunk = r->in + r->pos;
    r->pos += clen;
    return clen;
}

static int64_t ensure_internal(Reader* r, int64_t cap)
{
    if (r->internalCap < cap) {
        free(r->internal);
        r->internal = (uint8_t*)malloc((size_t)cap + 16);
        if (!r->internal) {
            r->internalCap = 0;
            return ACHIP_STATUS(ACHIP_CLASS_DEVICE, ACHIP_D_UNSUPPORTED);
        }
        r->internalCap = cap;
    }
    return 0;
}

/* Lz4HadoopInputStream.read(byte[], int, int) :61-82 ; single = read() :47-58 (dst unused, returns the byte) */
static int64_t lz4_read(Reader* r, uint8_t* dst, int64_t length, int single)
{
    while (r->chunkOff >= r->chunkLen) {
        const int64_t clen = next_chunk(r);
        if (clen < 0) {
            return clen;
        }
        int64_t beo = 0;
        if (!single && length >= r->blockLen) {  /* favor writing directly to the user buffer */
            const int64_t w = orc_lz4_decompress(r->chunk, clen, dst, length, &beo);
            if (w < 0) {
                r->eo = beo;
                return w;
            }
            r->chunkLen = w;
            r->chunkOff = w;
            return w;
        }
        const int64_t e = ensure_internal(r, (int64_t)r->bufferSize + 8);
        if (e < 0) {
            return e;
        }
        const int64_t w = orc_lz4_decompress(r->chunk, clen, r->internal, (int64_t)r->bufferSize + 8, &beo);
        if (w < 0) {
            r->eo = beo;
            return w;
        }
        r->chunkLen = w;
    }
    if (single) {
        return r->internal[r->chunkOff++];
    }
    const int64_t size = length < r->chunkLen - r->chunkOff ? length : r->chunkLen - r->chunkOff;
    memcpy(dst, r->internal + r->chunkOff, (size_t)size);
    r->chunkOff += size;
    return size;
}

/* SnappyHadoopInputStream.readNextChunk :91-141 ; returns 1 (decoded into the user's buffer), 0 (into the internal one, or nothing: chunkLen 0) or a status */
static int64_t snappy_next(Reader* r, uint8_t* user, int64_t ulen, int useInternal)
{
    const int64_t clen = next_chunk(r);
    if (clen == STREAM_EOF) {
        return 0;
    }
    if (clen < 0) {
        return clen;
    }
    int64_t beo = 0;
    const int64_t announced = orc_snappy_uncompressed_length(r->chunk, clen, &beo);
    if (announced < 0) {
        r->eo = beo;
        return announced;
    }
    r->chunkLen = announced;
    if (r->chunkLen > r->blockLen) {
        r->eo = r->pos - clen;
        return MALFORMED(ACHIP_D_HDP_CHUNK_EXCEEDS_BLOCK);
    }
    int direct = 1;
    uint8_t* target = user;
    int64_t cap = ulen;
    if (useInternal) {  /* read(): the "user buffer" is the internal one as it is */
        target = r->internal;
        cap = r->internalCap;
    }
    if (r->chunkLen > cap) {
        if (r->internalCap < r->chunkLen) {
            const int64_t e = ensure_internal(r, r->chunkLen + 8);
            if (e < 0) {
                return e;
            }
        }
        direct = 0;
        target = r->internal;
        cap = r->internalCap;
    }
    if (useInternal) {
        direct = 0;
    }
    uint8_t dummy[16];
    const int64_t w = orc_snappy_decompress(r->chunk, clen, target ? target : dummy, cap, &beo);
    if (w < 0) {
        r->eo = beo;
        return w;
    }
    if (w != r->chunkLen) {
        r->eo = r->pos - clen;
        return MALFORMED(ACHIP_D_HDP_LENGTH_MISMATCH);
    }
    return direct;
}

/* SnappyHadoopInputStream.read(byte[], int, int) :57-73 ; single = read() :44-54 */
static int64_t snappy_read(Reader* r, uint8_t* dst, int64_t length, int single)
{
    if (r->chunkOff >= r->chunkLen) {
        const int64_t direct = snappy_next(r, dst, length, single);
        if (direct < 0) {
            return direct;
        }
        if (r->chunkLen == 0) {
            return STREAM_EOF;
        }
        if (direct) {
            r->chunkOff += r->chunkLen;
            return r->chunkLen;
        }
    }
    if (single) {
        return r->internal[r->chunkOff++];
    }
    const int64_t size = length < r->chunkLen - r->chunkOff ? length : r->chunkLen - r->chunkOff;
    memcpy(dst, r->internal + r->chunkOff, (size_t)size);
    r->chunkOff += size;
    return size;
}

/* T/HadoopCodecDecompressor.java:40-60 over new XHadoopInputStream(decompressor, in[, bufferSize]) */
int64_t orc_hadoop_decompress(int32_t codec, const uint8_t* in, int64_t n, uint8_t* out, int64_t cap, int32_t bufferSize, int64_t* err_off)
{
    Reader r;
    memset(&r, 0, sizeof(r));
    r.codec = codec;
    r.bufferSize = bufferSize;
    r.in = in;
    r.n = n;
    int64_t done = 0;
    int64_t result = 0;
    while (done < cap) {
        const int64_t size = codec == 0 ? lz4_read(&r, out + done, cap - done, 0) : snappy_read(&r, out + done, cap - done, 0);
        if (size == STREAM_EOF) {
            break;
        }
        if (size < 0) {
            result = size;
            break;
        }
        done += size;
    }
    if (result == 0) {
        const int64_t b = codec == 0 ? lz4_read(&r, NULL, 0, 1) : snappy_read(&r, NULL, 0, 1);
        if (b >= 0) {
            r.eo = r.pos;
            result = ACHIP_STATUS(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_HDP_NOT_CONSUMED);
        }
        else if (b != STREAM_EOF) {
            result = b;
        }
    }
    free(r.internal);
    if (err_off) {
        *err_off = result < 0 ? r.eo : 0;
    }
    return result < 0 ? result : done;
}
