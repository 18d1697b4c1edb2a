/*
 * aircompressor_hip.h -- C ABI of libaircompressor_hip.so
 *
 * The MI355X (gfx950) batched block-codec backend for airlift/aircompressor's
 * LZ4 / Snappy / Zstd block API.  Every entry point is `extern "C"`, takes plain
 * pointers and integer sizes, and depends on nothing but libamdhip64.  This is
 * the boundary a Java `java.lang.foreign` (FFM) binding attaches to, exactly the
 * way the reference binds liblz4/libsnappy/libzstd:
 *
 *   reference binding mechanism ....... M/internal/NativeLoader.java:66-117
 *   reference LZ4 symbol record ....... M/lz4/Lz4Native.java:30-40
 *   reference Snappy symbol record .... M/snappy/SnappyNative.java:63-75
 *   reference Zstd symbol record ...... M/zstd/ZstdNative.java:27-41
 *   (M/ = src/main/java/io/airlift/compress/v3/ of airlift/aircompressor)
 *
 * FFM type map (NativeLoader.java:138-153): int8/int32/int64/pointer only --
 * every signature below uses only those.
 *
 * Return convention: >= 0 is "bytes written" (or a size); < 0 is an ACHIP status
 * (see achip_status_class / achip_status_detail).  Batched calls never abort a
 * batch: they fill status[i] / errOffset[i] per block.
 */
#ifndef AIRCOMPRESSOR_HIP_H
#define AIRCOMPRESSOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Status codes                                                              */
/* ------------------------------------------------------------------------- */
/* status = -(class + 16 * detail); class in 1..15, detail in 0..(2^26)       */
#define ACHIP_OK 0
#define ACHIP_CLASS_MALFORMED 1        /* -> MalformedInputException(offset, reason)  M/MalformedInputException.java:16-36 */
#define ACHIP_CLASS_OUTPUT_TOO_SMALL 2 /* -> IllegalArgumentException (buffer sizing) e.g. M/lz4/Lz4RawCompressor.java:87-89 */
#define ACHIP_CLASS_INVALID_ARGUMENT 3 /* -> IllegalArgumentException (bad args)      */
#define ACHIP_CLASS_DEVICE 4           /* HIP runtime / device failure                */

#define ACHIP_STATUS(cls, detail) (-((cls) + 16 * (detail)))

/* detail ids: one per distinct message of the reference's Java codecs. */
enum achip_detail {
    ACHIP_D_GENERIC = 0,
    /* LZ4 decode -- M/lz4/Lz4RawDecompressor.java */
    ACHIP_D_LZ4_INPUT_EMPTY = 1,          /* :48-50  "input is empty"                                  */
    ACHIP_D_LZ4_MALFORMED = 2,            /* :66,76,125,138 "Malformed input"                          */
    ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE = 3, /* :85   "attempt to write last literal outside of destination buffer" */
    ACHIP_D_LZ4_INPUT_NOT_CONSUMED = 4,   /* :89   "all input must be consumed"                        */
    ACHIP_D_LZ4_OFFSET_OUTSIDE = 5,       /* :118  "offset outside destination buffer"                 */
    ACHIP_D_LZ4_LAST_5_LITERALS = 6,      /* :170  "last 5 bytes must be literals"                     */
    ACHIP_D_LZ4_EMPTY_OUTPUT = 7,         /* :52-57 the reference *returns -1* here (no exception)     */
    /* LZ4 encode -- M/lz4/Lz4RawCompressor.java */
    ACHIP_D_LZ4_MAX_INPUT = 8,            /* :83-85 "Max input length exceeded"                        */
    ACHIP_D_LZ4_MAX_OUTPUT = 9,           /* :87-89 "Max output length must be larger than N"          */
    /* Snappy decode -- M/snappy/SnappyRawDecompressor.java */
    ACHIP_D_SNAPPY_MALFORMED = 16,        /* :92,108,119,128,155,160,164 "Malformed input"             */
    ACHIP_D_SNAPPY_TRUNCATED = 17,        /* :316-318 "Input is truncated"                             */
    ACHIP_D_SNAPPY_LEN_HIGH_BIT = 18,     /* :300   "last byte of compressed length int has high bit set" */
    ACHIP_D_SNAPPY_INVALID_LENGTH = 19,   /* :308   "invalid compressed length"                        */
    ACHIP_D_SNAPPY_LENGTH_MISMATCH = 20,  /* :61-65 "Recorded length is N bytes but actual length ..." */
    ACHIP_D_SNAPPY_OUTPUT_TOO_SMALL = 21, /* :49-50 "Uncompressed length N must be less than M"        */
    /* Snappy encode -- M/snappy/SnappyRawCompressor.java */
    ACHIP_D_SNAPPY_MAX_OUTPUT = 22,       /* :85-88 "Output buffer must be at least N bytes"           */
    /* Zstd decode -- M/zstd/ZstdFrameDecompressor.java and friends */
    ACHIP_D_ZSTD_NOT_ENOUGH_INPUT = 32,   /* "Not enough input bytes"                                  */
    ACHIP_D_ZSTD_OUTPUT_TOO_SMALL = 33,   /* "Output buffer too small" (verify => MalformedInputException) */
    ACHIP_D_ZSTD_CORRUPTED = 34,          /* "Input is corrupted"                                      */
    ACHIP_D_ZSTD_BAD_MAGIC = 35,          /* :949-959 "Invalid magic prefix: <hex>"                    */
    ACHIP_D_ZSTD_V07_MAGIC = 36,          /* :955   "Data encoded in unsupported ZSTD v0.7 format"     */
    ACHIP_D_ZSTD_BAD_CHECKSUM = 37,       /* :199-201 "Bad checksum. Expected: .., actual: .."         */
    ACHIP_D_ZSTD_DICTIONARY = 38,         /* :905   "Custom dictionaries not supported"                */
    ACHIP_D_ZSTD_INVALID_BLOCK_TYPE = 39, /* :186   "Invalid block type"                               */
    ACHIP_D_ZSTD_BLOCK_TOO_LARGE = 40,    /* :278   (blockSize > 128 KiB)                              */
    ACHIP_D_ZSTD_BLOCK_TOO_SMALL = 41,    /* :279   "Compressed block size too small"                  */
    ACHIP_D_ZSTD_WINDOW_TOO_LARGE = 42,   /* :303   "Window size too large (not yet supported)"        */
    ACHIP_D_ZSTD_DICT_CORRUPTED = 43,     /* :294   treeless literals without a table                  */
    ACHIP_D_ZSTD_LITERALS_TOO_LARGE = 44, /* :752,795 "Block exceeds maximum size"                     */
    ACHIP_D_ZSTD_FSE_TABLE_LOG = 45,      /* FseTableReader.java:46 "FSE table size exceeds maximum allowed size" */
    ACHIP_D_ZSTD_FSE_SYMBOL = 46,         /* FseTableReader.java:74,127 symbol too large               */
    ACHIP_D_ZSTD_TABLE_MISSING = 47,      /* :622,645,668 repeat mode without a previous table         */
    ACHIP_D_ZSTD_VALUE_TOO_LARGE = 48,    /* :616,639,662 "Value exceeds expected maximum value"       */
    ACHIP_D_ZSTD_BITSTREAM_EMPTY = 49,    /* BitInputStream.java:112 "Bitstream is empty"              */
    ACHIP_D_ZSTD_BITSTREAM_NO_MARK = 50,  /* BitInputStream.java:115 "Bitstream end mark not present"  */
    ACHIP_D_ZSTD_BITSTREAM_NOT_CONSUMED = 51, /* Huffman.java:316 "Bit stream is not fully consumed"   */
    ACHIP_D_ZSTD_SEQUENCES_NOT_CONSUMED = 52, /* :396 "Not all sequences were consumed"                */
    ACHIP_D_ZSTD_FSE_OUTPUT_SMALL = 53,   /* FiniteStateEntropy.java:114,131 "Output buffer is too small" */
    /* Zstd encode */
    ACHIP_D_ZSTD_MAX_OUTPUT = 60,         /* Util.checkArgument "Output buffer too small"              */
    /* LZ4 frame container -- M/lz4/Lz4FrameCompression.java */
    ACHIP_D_LZ4F_TOO_SHORT = 64,          /* :150   "Input is too short to be an LZ4 frame"                 */
    ACHIP_D_LZ4F_TRUNC_MAGIC = 65,        /* :158   "Truncated LZ4 frame: incomplete magic number"          */
    ACHIP_D_LZ4F_BAD_MAGIC = 66,          /* :170   "Invalid LZ4 frame magic number"                        */
    ACHIP_D_LZ4F_TRUNC_HEADER = 67,       /* :191,230 "Truncated LZ4 frame header"                          */
    ACHIP_D_LZ4F_VERSION_0 = 68,          /* :198   "Unsupported LZ4 frame version: 0" (2, 3 follow)        */
    ACHIP_D_LZ4F_VERSION_2 = 69,
    ACHIP_D_LZ4F_VERSION_3 = 70,
    ACHIP_D_LZ4F_RESERVED_BITS = 71,      /* :202   "Corrupt LZ4 frame: reserved bits in the frame descriptor must be zero" */
    ACHIP_D_LZ4F_LINKED_BLOCKS = 72,      /* :213   "LZ4 frames with linked blocks are not supported"       */
    ACHIP_D_LZ4F_DICTIONARY = 73,         /* :217   "LZ4 frames with a dictionary are not supported"        */
    ACHIP_D_LZ4F_BLOCK_MAX_SIZE = 74,     /* :222   "Invalid LZ4 frame block maximum size"                  */
    ACHIP_D_LZ4F_HEADER_CHECKSUM = 75,    /* :241   "Corrupt LZ4 frame: invalid header checksum"            */
    ACHIP_D_LZ4F_MISSING_BLOCK_SIZE = 76, /* :248   "Truncated LZ4 frame: missing block size"               */
    ACHIP_D_LZ4F_BLOCK_PAST_END = 77,     /* :260   "Truncated LZ4 frame: block extends past end of input"  */
    ACHIP_D_LZ4F_OUTPUT_TOO_SMALL = 78,   /* :265,276 "Output buffer too small" (a MalformedInputException here)  */
    ACHIP_D_LZ4F_BLOCK_EXCEEDS_MAX = 79,  /* :279   "Corrupt LZ4 frame: decompressed block exceeds maximum block size" */
    ACHIP_D_LZ4F_MISSING_BLOCK_CHECKSUM = 80, /* :288 "Truncated LZ4 frame: missing block checksum"         */
    ACHIP_D_LZ4F_BLOCK_CHECKSUM = 81,     /* :293   "Corrupt LZ4 frame: invalid block checksum"             */
    ACHIP_D_LZ4F_MISSING_CONTENT_CHECKSUM = 82, /* :307 "Truncated LZ4 frame: missing content checksum"     */
    ACHIP_D_LZ4F_CONTENT_CHECKSUM = 83,   /* :312   "Corrupt LZ4 frame: invalid content checksum"           */
    ACHIP_D_LZ4F_CONTENT_SIZE = 84,       /* :318   "Corrupt LZ4 frame: content size does not match frame header" */
    ACHIP_D_LZ4F_TRUNC_SKIP_SIZE = 85,    /* :334   "Truncated LZ4 skippable frame: missing frame size"     */
    ACHIP_D_LZ4F_TRUNC_SKIP = 86,         /* :340   "Truncated LZ4 skippable frame"                         */
    ACHIP_D_LZ4F_MAX_OUTPUT = 87,         /* :362   "Output buffer too small" (IllegalArgumentException, encoder) */
    /* x-snappy-framed streams (M/snappy/SnappyFramedInputStream.java; IOException / EOFException unless noted) */
    ACHIP_D_SNF_EOF_STREAM_HEADER = 88,   /* :68    "encountered EOF while reading stream header"           */
    ACHIP_D_SNF_BAD_STREAM_HEADER = 89,   /* :71    "invalid stream header"                                 */
    ACHIP_D_SNF_EOF_BLOCK_HEADER = 90,    /* :323   "encountered EOF while reading block header"            */
    ACHIP_D_SNF_EOF_FRAME = 91,           /* :177   "unexpectd EOF when reading frame" (sic)                */
    ACHIP_D_SNF_STREAM_ID_LENGTH = 92,    /* :255   "stream identifier chunk with invalid length: N"        */
    ACHIP_D_SNF_UNSKIPPABLE = 93,         /* :263   "unsupported unskippable chunk: XX"                     */
    ACHIP_D_SNF_INVALID_LENGTH = 94,      /* :273   "invalid length: N for chunk flag: XX"                  */
    ACHIP_D_SNF_CHECKSUM = 95,            /* :207   "Corrupt input: invalid checksum"                       */
    ACHIP_D_SNF_OUTPUT_TOO_SMALL = 96,    /* this API: the destination cannot hold the stream's plaintext   */
    ACHIP_D_SNF_MAX_OUTPUT = 97,          /* this API (encoder): dstCap < achip_snappyframed_max_compressed_length */
    /* Hadoop LZ4 / Snappy block streams (M/lz4/Lz4HadoopInputStream.java, M/snappy/SnappyHadoopInputStream.java; IOException / EOFException) */
    ACHIP_D_HDP_TRUNCATED_INT = 104,      /* Lz4HadoopInputStream.java:153 "Stream is truncated"                                     */
    ACHIP_D_HDP_EOF_BLOCK_DATA = 105,     /* :136   "encountered EOF while reading block data"                                        */
    ACHIP_D_HDP_CHUNK_EXCEEDS_BLOCK = 106, /* SnappyHadoopInputStream.java:117 "Chunk uncompressed size is greater than block size"   */
    ACHIP_D_HDP_LENGTH_MISMATCH = 107,    /* SnappyHadoopInputStream.java:137 "Expected to read N bytes, but data only contained M bytes" */
    ACHIP_D_HDP_NOT_CONSUMED = 108,       /* T/HadoopCodecDecompressor.java:52 "All input was not consumed": the destination cannot hold the stream */
    ACHIP_D_HDP_NEGATIVE_LENGTH = 109,    /* this API: a negative chunk length (Java: the block codec's range check throws)           */
    ACHIP_D_HDP_MAX_OUTPUT = 110,         /* this API (encoder): dstCap < achip_hadoop_max_compressed_length                          */
    /* runtime */
    ACHIP_D_NO_DEVICE = 100,
    ACHIP_D_HIP_ERROR = 101,
    ACHIP_D_BAD_ARGUMENT = 102,
    ACHIP_D_UNSUPPORTED = 103
};

int32_t achip_status_class(int32_t status);  /* 0 for status >= 0 */
int32_t achip_status_detail(int32_t status);
/* English reason string for a detail id, matching the reference's exception text
 * up to the ": offset=N" suffix that MalformedInputException appends itself. */
const char* achip_detail_message(int32_t detail);

/* ------------------------------------------------------------------------- */
/* Library / device                                                          */
/* ------------------------------------------------------------------------- */
const char* achip_version(void);
int32_t achip_device_count(void); /* 0 when no HIP device is usable; isEnabled() keys off this, cf. M/lz4/Lz4Native.java:75-85 */
const char* achip_last_error(void); /* thread-local text of the last ACHIP_CLASS_DEVICE failure */

/* ------------------------------------------------------------------------- */
/* Size helpers (host only, no device)                                        */
/* ------------------------------------------------------------------------- */
/* replaces Lz4RawCompressor.maxCompressedLength      M/lz4/Lz4RawCompressor.java:64-67 (bound as LZ4_compressBound in M/lz4/Lz4Native.java:31) */
int32_t achip_lz4_max_compressed_length(int32_t uncompressedSize);
/* replaces SnappyRawCompressor.maxCompressedLength   M/snappy/SnappyRawCompressor.java:47-70 (snappy_max_compressed_length, M/snappy/SnappyNative.java:68) */
int32_t achip_snappy_max_compressed_length(int32_t uncompressedSize);
/* replaces ZstdJavaCompressor.maxCompressedLength    M/zstd/ZstdJavaCompressor.java:31-40 (ZSTD_compressBound, M/zstd/ZstdNative.java:28) */
int32_t achip_zstd_max_compressed_length(int32_t uncompressedSize);
/* replaces SnappyRawDecompressor.getUncompressedLength  M/snappy/SnappyRawDecompressor.java:30-33,277-321; negative = status, *errOffset set */
int64_t achip_snappy_uncompressed_length(const void* src, int64_t srcLen, int64_t* errOffset);
/* replaces ZstdFrameDecompressor.getDecompressedSize    M/zstd/ZstdFrameDecompressor.java:942-947; -1 = unknown, < -1 = status */
int64_t achip_zstd_decompressed_size(const void* src, int64_t srcLen, int64_t* errOffset);
/* what the reading side of the stream classes needs in one-shot form (M/zstd/ZstdInputStream.java:63-105 over
 * M/zstd/ZstdIncrementalFrameDecompressor.java:99-352 reads frames WITHOUT a content size -- ZstdOutputStream's from 4 MiB on -- through a
 * growing window): an upper bound of the decoded size of all frames in the buffer, from the frame and block headers alone (host code);
 * >= 0 bound, negative = status, *errOffset set */
int64_t achip_zstd_decompress_bound(const void* src, int64_t srcLen, int64_t* errOffset);

/* ------------------------------------------------------------------------- */
/* Context: one HIP stream + device scratch on one GPU.                       */
/* Not thread-safe (like one codec instance, M/lz4/Lz4JavaCompressor.java:27-29);*/
/* distinct contexts may be used concurrently from distinct threads.           */
/* ------------------------------------------------------------------------- */
typedef struct achip_ctx achip_ctx;

achip_ctx* achip_ctx_create(int32_t device);      /* NULL on failure; see achip_last_error */
void achip_ctx_destroy(achip_ctx* ctx);
int32_t achip_ctx_device(achip_ctx* ctx);
void* achip_ctx_stream(achip_ctx* ctx);           /* the hipStream_t the batched calls launch on */
int32_t achip_ctx_synchronize(achip_ctx* ctx);    /* hipStreamSynchronize */
/* tuning knobs (kernel variant selection); name/value pairs documented in DESIGN.md. returns 0 or status */
int32_t achip_ctx_set_option(achip_ctx* ctx, const char* name, int64_t value);
/* diagnostics of the LAST batched call on this context (synchronizes the stream); -1 = unknown name / nothing recorded.
 * "zstd.decompress.fallback_items": items the five-stage pipeline handed to the one-kernel decoder;
 * "zstd.decompress.fallback_stage1" .. "stage6": the same, by the stage that handed them over (6: the multi-block stages' walk);
 * "zstd.decompress.multiblock_items" / "_blocks" / "_fast_items": items of the last Zstd decode that hold one frame of several blocks
 *   (ZstdOutputStream's output; ZstdFrameCompressor's and libzstd's beyond 128 KiB), their blocks, and how many of them the pipeline's
 *   multi-block stages finished (the rest went to the one-kernel decoder); option "zstd.decompress.stream_blocks" sizes those stages;
 * "lz4.decompress.mixed_groups": auto mode's count of mixed 16-block groups of the last LZ4 / Snappy decode (-1: no probe ran);
 * "decompress.choice": the decoder auto mode ran (0 LDS rings, 1 a lane per block with copy steps, 2 a lane per block with an LDS
 *   output window; -1: no probe ran).  DESIGN.md 8b lists every option and statistic. */
int64_t achip_ctx_get_stat(achip_ctx* ctx, const char* name);

/* device / pinned memory helpers; addresses are usable as MemorySegment.ofAddress */
void* achip_device_alloc(achip_ctx* ctx, int64_t bytes);
int32_t achip_device_free(achip_ctx* ctx, void* p);
void* achip_host_alloc_pinned(int64_t bytes);
int32_t achip_host_free_pinned(void* p);
int32_t achip_memcpy_h2d(achip_ctx* ctx, void* dst, const void* src, int64_t bytes); /* async on the ctx stream */
int32_t achip_memcpy_d2h(achip_ctx* ctx, void* dst, const void* src, int64_t bytes); /* async on the ctx stream */
int32_t achip_memset_d(achip_ctx* ctx, void* dst, int32_t value, int64_t bytes);     /* async on the ctx stream */

/* HIP events on the ctx stream, for timing from a host language */
void* achip_event_create(void);
int32_t achip_event_destroy(void* ev);
int32_t achip_event_record(achip_ctx* ctx, void* ev);
float achip_event_elapsed_ms(void* evStart, void* evStop); /* synchronizes on evStop; <0 on failure */

/* ------------------------------------------------------------------------- */
/* Batched, device-resident block API (the hot path).                         */
/*                                                                           */
/* Every pointer is a DEVICE-accessible address (device memory, or pinned     */
/* host memory).  Block i reads srcBase[srcOff[i] .. +srcLen[i]) and writes   */
/* dstBase[dstOff[i] .. +dstCap[i]); on return (after achip_ctx_synchronize)  */
/* outLen[i] = bytes written, status[i] = 0 or an ACHIP status,               */
/* errOffset[i] = the offset the reference's exception would carry.           */
/* The call itself is asynchronous on the ctx stream; its return value only   */
/* reports launch failures.  Regions of distinct blocks must not overlap --    */
/* the WHOLE capacity [dstOff[i], dstOff[i] + dstCap[i]) belongs to block i    */
/* for the call: as with the Java decoders' 8-byte copies, bytes between      */
/* outLen[i] and dstCap[i] may be written (their content is unspecified).     */
/* ------------------------------------------------------------------------- */
#define ACHIP_BATCH_ARGS                                                            \
    achip_ctx *ctx, const void *srcBase, const int64_t *srcOff, const int32_t *srcLen, \
        void *dstBase, const int64_t *dstOff, const int32_t *dstCap, int32_t *outLen, \
        int32_t *status, int64_t *errOffset, int32_t nBlocks

/* replaces Lz4RawDecompressor.decompress    M/lz4/Lz4RawDecompressor.java:35-198   (a1) */
int32_t achip_lz4_decompress_batch(ACHIP_BATCH_ARGS);
/* replaces Lz4RawCompressor.compress        M/lz4/Lz4RawCompressor.java:69-192     (a2) */
int32_t achip_lz4_compress_batch(ACHIP_BATCH_ARGS);
/* replaces SnappyRawDecompressor.decompress M/snappy/SnappyRawDecompressor.java:35-220 (a3) */
int32_t achip_snappy_decompress_batch(ACHIP_BATCH_ARGS);
/* replaces SnappyRawCompressor.compress     M/snappy/SnappyRawCompressor.java:74-233   (a4) */
int32_t achip_snappy_compress_batch(ACHIP_BATCH_ARGS);
/* replaces ZstdFrameDecompressor.decompress M/zstd/ZstdFrameDecompressor.java:135-210  (a5-a10); frames of several blocks -- what
 * ZstdOutputStream (M/zstd/ZstdOutputStream.java:154-221) writes -- take the same fast path (f3, decode half) */
int32_t achip_zstd_decompress_batch(ACHIP_BATCH_ARGS);
/* replaces ZstdFrameCompressor.compress(level 3) M/zstd/ZstdFrameCompressor.java:136-150 (a11-a14) */
int32_t achip_zstd_compress_batch(ACHIP_BATCH_ARGS);

/* ------------------------------------------------------------------------- */
/* Single-block host-pointer API: what Compressor.compress(MemorySegment,     */
/* MemorySegment) / Decompressor.decompress(...) bind to (M/Compressor.java:  */
/* 20-35, M/Decompressor.java:23-30).  src/dst are HOST pointers; the call    */
/* stages through the context's pinned buffers, runs a 1-block batch and      */
/* synchronizes.  Argument order follows LZ4_compress_fast /                  */
/* LZ4_decompress_safe as bound in M/lz4/Lz4Native.java:33,37.                */
/* ------------------------------------------------------------------------- */
int32_t achip_lz4_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_lz4_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_snappy_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_snappy_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_zstd_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_zstd_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);

/* ---- LZ4 frame container ----
 * Replace Lz4FrameCompression.maxCompressedLength / compress / decompress    M/lz4/Lz4FrameCompression.java:70-83, 96-140, 145-343
 * (the methods behind Lz4FrameJavaCompressor / Lz4FrameJavaDecompressor, M/lz4/Lz4FrameJavaCompressor.java:26-44) with the
 * HIP block codec underneath.  An item of the batch is a whole buffer: any number of concatenated and skippable frames on
 * decode, one frame (4 MiB independent blocks, no checksums) on encode.  Same batch arguments, result convention and
 * asynchrony as the block codecs; errOffset carries the MalformedInputException offset. */
int32_t achip_lz4frame_max_compressed_length(int32_t uncompressedSize);  /* negative: IllegalArgumentException */
int32_t achip_lz4frame_decompress_batch(ACHIP_BATCH_ARGS);
int32_t achip_lz4frame_compress_batch(ACHIP_BATCH_ARGS);
/* one HOST buffer, staged through the context's pinned buffer (Lz4FrameJava{De,}Compressor.{de,}compress(byte[]...)) */
int32_t achip_lz4frame_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_lz4frame_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);

/* ---- x-snappy-framed streams ----
 * Replace SnappyFramedOutputStream used as "write everything, close"   M/snappy/SnappyFramedOutputStream.java:73-96, 113-145, 200-255
 *     and SnappyFramedInputStream read to the end of the stream       M/snappy/SnappyFramedInputStream.java:52-73, 135-305
 * (64 KiB blocks, masked CRC-32C of every block's plaintext  M/snappy/Crc32C.java:29-50, a block stored raw when it does not
 * reach 0.85) with the HIP block codec underneath.  An item of the batch is a whole stream.  Same batch arguments, result
 * convention and asynchrony as the block codecs.  The stream-level IOExceptions (details ACHIP_D_SNF_*) report the position of
 * the offending chunk header in errOffset; a block codec error keeps its own detail and offset. */
int32_t achip_snappyframed_max_compressed_length(int32_t uncompressedSize);  /* bound this API asks of dstCap; negative: IllegalArgumentException */
int32_t achip_snappyframed_decompress_batch(ACHIP_BATCH_ARGS);
int32_t achip_snappyframed_compress_batch(ACHIP_BATCH_ARGS);
/* one HOST buffer, staged through the context's pinned buffer */
int32_t achip_snappyframed_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_snappyframed_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);

/* ---- Hadoop LZ4 / Snappy block streams (second half) ----
 * Replace Lz4HadoopOutputStream / SnappyHadoopOutputStream used as "write everything, close"   M/lz4/Lz4HadoopOutputStream.java:60-118,
 *     M/snappy/SnappyHadoopOutputStream.java:60-131   (as T/HadoopCodecCompressor.java:57-72 drives them)
 * and Lz4HadoopInputStream / SnappyHadoopInputStream read to the end   M/lz4/Lz4HadoopInputStream.java:47-156,
 *     M/snappy/SnappyHadoopInputStream.java:44-170   (as T/HadoopCodecDecompressor.java:40-60 drives them: a destination that cannot hold
 *     the stream is ACHIP_D_HDP_NOT_CONSUMED)
 * -- the streams behind org.apache.hadoop.io.compress.Lz4Codec / SnappyCodec (M/lz4/Lz4HadoopStreams.java:52-66) -- with the HIP block
 * codecs underneath.  [BE int plaintext length][BE int compressed length][block] ...; an item of the batch is a whole stream.  The
 * streams' buffer size (256 KiB unless configured, M/lz4/Lz4HadoopStreams.java:30) is the context option "hadoop.buffer_size".  Same
 * batch arguments, result convention and asynchrony as the block codecs; stream-level IOExceptions (details ACHIP_D_HDP_*) report the
 * position where the failing read began, a block codec error keeps its own detail and offset. */
int32_t achip_hadoop_max_compressed_length(int32_t codec /* 0 LZ4, 1 Snappy */, int32_t uncompressedSize, int32_t bufferSize);  /* bound this API asks of dstCap */
int32_t achip_lz4hadoop_decompress_batch(ACHIP_BATCH_ARGS);
int32_t achip_lz4hadoop_compress_batch(ACHIP_BATCH_ARGS);
int32_t achip_snappyhadoop_decompress_batch(ACHIP_BATCH_ARGS);
int32_t achip_snappyhadoop_compress_batch(ACHIP_BATCH_ARGS);
/* one HOST buffer, staged through the context's pinned buffer */
int32_t achip_lz4hadoop_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_lz4hadoop_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_snappyhadoop_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);
int32_t achip_snappyhadoop_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);

/* ---- Zstd streams ----
 * READING needs no entry point of its own: what ZstdInputStream (M/zstd/ZstdInputStream.java:26-151, through
 * ZstdIncrementalFrameDecompressor.java:44-72,216-234) yields for a stream is what ZstdFrameDecompressor yields for its bytes, and
 * achip_zstd_decompress* take frames of any number of blocks on their fast path (DESIGN 6 "Multi-block frames").
 * WRITING: item i = everything a caller hands to ONE ZstdOutputStream (M/zstd/ZstdOutputStream.java:30-221) -- write(buffer, 0, n) and
 * close(), the way the reference's stream harness drives it (T/HadoopCodecCompressor.java:57-72); dst receives what the stream puts on
 * its sink.  That is NOT achip_zstd_compress' output: the stream's parameters are those for an unknown input size (window 2^20, hash
 * 2^17, chain 2^16 whatever n is; :48-58).  Below 4 MiB close() writes the input as one chunk whose size the frame header announces; from
 * 4 MiB on the stream flushes chunks before close() (header without content size, 23 blocks, then 15 per further 1920 KiB, the window slid
 * in between -- and, as in the reference, 7 blocks after every slide without a match: DESIGN 10 row 3).  n >= 2^30 is INVALID_ARGUMENT /
 * ACHIP_D_UNSUPPORTED (the Java code overflows there); option "zstd.stream.chunked" = 0 refuses everything from 4 MiB on the same way.
 * Byte-identical output wants dstCap >= achip_zstdstream_max_compressed_length(n) (the stream itself never runs out of room). */
int32_t achip_zstdstream_max_compressed_length(int32_t uncompressedSize);
int32_t achip_zstdstream_compress_batch(ACHIP_BATCH_ARGS);
int32_t achip_zstdstream_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset);

/* ---- xxhash: batched XXH64 / XXH32 of device-resident buffers ----
 * Replace XxHash64Hasher.hash(MemorySegment input, long seed)   M/xxhash/XxHash64Hasher.java:78-86  (-> XxHash64JavaHasher.java:126)
 *     and XxHash32Hasher.hash(MemorySegment input, int seed)    M/xxhash/XxHash32Hasher.java       (-> XxHash32JavaHasher.java:112)
 * for nBuffers buffers per call: buffer i = srcBase + srcOff[i], srcLen[i] bytes; outHash[i] receives the hash
 * (all arrays device memory; asynchronous on the context's stream).  Returns 0 or a negative status. */
int32_t achip_xxhash64_batch(achip_ctx* ctx, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, int64_t seed, int64_t* outHash, int32_t nBuffers);
int32_t achip_xxhash32_batch(achip_ctx* ctx, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, int32_t seed, int32_t* outHash, int32_t nBuffers);
/* one HOST buffer (staged through the context's pinned buffer; synchronous): the one-shot form of the same two methods */
int32_t achip_xxhash64(achip_ctx* ctx, const void* src, int64_t srcLen, int64_t seed, int64_t* outHash);
int32_t achip_xxhash32(achip_ctx* ctx, const void* src, int64_t srcLen, int32_t seed, int32_t* outHash);

/* Host-pointer batch: nBlocks independent blocks described by HOST arrays of HOST
 * pointers' offsets relative to srcBase/dstBase (host).  Stages in, launches the
 * device batch for `codecOp`, stages out, synchronizes.  codecOp: see below. */
#define ACHIP_OP_LZ4_DECOMPRESS 0
#define ACHIP_OP_LZ4_COMPRESS 1
#define ACHIP_OP_SNAPPY_DECOMPRESS 2
#define ACHIP_OP_SNAPPY_COMPRESS 3
#define ACHIP_OP_ZSTD_DECOMPRESS 4
#define ACHIP_OP_ZSTD_COMPRESS 5
#define ACHIP_OP_LZ4FRAME_DECOMPRESS 6
#define ACHIP_OP_LZ4FRAME_COMPRESS 7
#define ACHIP_OP_SNAPPYFRAMED_DECOMPRESS 8
#define ACHIP_OP_SNAPPYFRAMED_COMPRESS 9
#define ACHIP_OP_LZ4HADOOP_DECOMPRESS 10
#define ACHIP_OP_LZ4HADOOP_COMPRESS 11
#define ACHIP_OP_SNAPPYHADOOP_DECOMPRESS 12
#define ACHIP_OP_SNAPPYHADOOP_COMPRESS 13
#define ACHIP_OP_ZSTDSTREAM_COMPRESS 14
int32_t achip_batch_host(int32_t codecOp, ACHIP_BATCH_ARGS);
/* (Staging is chunked and pipelined over up to four pinned slots: this thread gathers chunk c+1 into pinned memory while chunk c
 * is uploaded / run / downloaded on three streams and a finalizer thread scatters chunk c-1 to dstBase -- each side with a few
 * copy threads of its own; one upload and one download per chunk.  A single chunk (e.g. a single block) runs in order on the
 * context stream.  Options "host.chunk_bytes", "host.copy_threads".  Replaces N calls of Compressor.compress(byte[]...) /
 * Decompressor.decompress(byte[]...), M/Compressor.java:20-35, M/Decompressor.java:23-30, over one device.) */

/* Mixed batches (BASELINE configs[4]): item i is processed by codecOp[i] (ACHIP_OP_*), any interleaving.  The items are
 * bucketed by codec op so that every kernel launch is homogeneous -- what a caller holding e.g. one ORC / Parquet stripe with
 * LZ4, Snappy and Zstd pages would otherwise do by hand with one Decompressor per codec (M/Decompressor.java:23-30) -- and the
 * results land in the caller's item order.
 *   achip_mixed_batch:      codecOp is a HOST array (the caller's own knowledge of its items); every other array and both buffers are
 *                           DEVICE-accessible as for the homogeneous batch calls; asynchronous on the ctx stream (the buckets of the
 *                           Snappy and Zstd families run on streams of the library's own, side by side with LZ4's, ordered by events behind
 *                           whatever the ctx stream held and in front of whatever is enqueued on it next: option "mixed.concurrent").
 *   achip_mixed_batch_host: everything is HOST memory; staged like achip_batch_host; synchronous. */
int32_t achip_mixed_batch(achip_ctx* ctx, const int32_t* codecOp, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, void* dstBase,
                          const int64_t* dstOff, const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t nBlocks);
int32_t achip_mixed_batch_host(achip_ctx* ctx, const int32_t* codecOp, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, void* dstBase,
                               const int64_t* dstOff, const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t nBlocks);

/* One process, several devices: the batch is cut into nCtx contiguous slices balanced by srcLen[i] + dstCap[i] (the rule of
 * achip_partition_blocks) and slice d runs through achip_batch_host (codecOps == NULL: every item is `codecOp`) or
 * achip_mixed_batch_host (codecOps[i] per item) on ctxs[d], each slice in a host thread of its own, all inside this call -- what
 * a JVM (one process) does with the GPUs of a node; java/.../HipBatchCodec.run is the same split with Java threads.  ctxs: nCtx
 * distinct contexts (normally one per device; several on one device are legal).  sliceStarts (may be NULL) receives the nCtx + 1
 * slice boundaries.  Units are self-contained (each frame decode resets its state, M/zstd/ZstdFrameDecompressor.java:151; each
 * compress() builds a fresh context, M/zstd/ZstdFrameCompressor.java:162; LZ4 / Snappy blocks likewise): the slices
 * exchange nothing.  Returns 0, or the first failing slice's negative status (achip_last_error names the slice). */
int32_t achip_multi_batch_host(achip_ctx* const* ctxs, int32_t nCtx, int32_t codecOp, const int32_t* codecOps, const void* srcBase, const int64_t* srcOff,
                               const int32_t* srcLen, void* dstBase, const int64_t* dstOff, const int32_t* dstCap, int32_t* outLen, int32_t* status,
                               int64_t* errOffset, int32_t nBlocks, int32_t* sliceStarts);

/* One long Zstd stream in bounded memory: what ZstdInputStream does over ZstdIncrementalFrameDecompressor
 * (M/zstd/ZstdInputStream.java:63-105, M/zstd/ZstdIncrementalFrameDecompressor.java:44-72,216-234).  Input arrives in pieces, output
 * leaves in pieces, a frame may be any length (ZstdOutputStream writes ONE frame per stream), frames may follow each other.
 *   begin   a stream state on ctx (NULL: achip_last_error); per state ~45 MB of host + device memory at an 8 MiB window, whatever
 *           the stream's length
 *   feed    takes up to srcLen bytes of src (*consumed; it holds back at most one step of blocks), writes up to dstCap bytes to dst
 *           (*produced).  Returns 0, or a negative status once the stream is known to be damaged AND every byte in front of the damage
 *           has been delivered -- the call that would deliver the first byte of a damaged block fails, where ZstdInputStream.read
 *           throws (*errOffset: stream offset of the block).  *consumed < srcLen with *produced == dstCap: output room needed;
 *           *consumed == srcLen and *produced < dstCap: input needed (or the stream's end: at_stopping_point)
 *   at_stopping_point   1 when a frame has been read to its end, its bytes are delivered and the next frame's magic is not complete
 *           (ZstdIncrementalFrameDecompressor.isAtStoppingPoint: state READ_FRAME_MAGIC -- up to three bytes behind the last frame are
 *           ignored, as ZstdInputStream.java:81-86 ignores them; before the first frame the Java state is INITIAL and this returns 0):
 *           where a stream may end; the end of input anywhere else is "Not enough input bytes" (ZstdInputStream.java:86).
 *           (RAW / RLE blocks that say more than 128 KiB -- the format forbids them, the Java reader copies / fills what they say,
 *           ZstdIncrementalFrameDecompressor.java:204-226 -- are decoded as several blocks of at most 128 KiB.)
 *   end     releases the state
 * All host pointers; synchronous; a state serves one thread at a time like its context. */
void* achip_zstdstream_decompress_begin(achip_ctx* ctx);
int32_t achip_zstdstream_decompress_feed(achip_ctx* ctx, void* state, const void* src, int64_t srcLen, void* dst, int64_t dstCap, int64_t* consumed, int64_t* produced,
                                         int64_t* errOffset);
int32_t achip_zstdstream_decompress_at_stopping_point(void* state);
int32_t achip_zstdstream_decompress_end(achip_ctx* ctx, void* state);

/* ... and the writer: ZstdOutputStream (M/zstd/ZstdOutputStream.java:93-221) a chunk at a time, in the 4 MiB the Java stream buffers -- write()
 * appends to the stream's buffer (on the device), a full buffer is flushed as whole blocks (compressIfNecessary :122-131, writeChunk :154-221: the
 * window slides), close() writes the rest and the checksum.  The bytes are the Java stream's whatever the sizes of the calls.
 *   begin    a stream state on ctx (NULL: achip_last_error)
 *   feed     write(src, 0, srcLen): takes input until dst is full of flushed blocks (*consumed, *produced); 0 or a negative status
 *   finish   close(): 1 when the stream's last byte has been delivered, 0 when dst was too small for the rest (call again)
 *   end      releases the state */
void* achip_zstdstream_compress_begin(achip_ctx* ctx);
int32_t achip_zstdstream_compress_feed(achip_ctx* ctx, void* state, const void* src, int64_t srcLen, void* dst, int64_t dstCap, int64_t* consumed, int64_t* produced);
int32_t achip_zstdstream_compress_finish(achip_ctx* ctx, void* state, void* dst, int64_t dstCap, int64_t* produced);
int32_t achip_zstdstream_compress_end(achip_ctx* ctx, void* state);

/* Balanced contiguous partition of a batch over nParts GPUs: fills
 * starts[0..nParts] with block indices so that each part's sum of weight[i] is
 * as equal as a contiguous split allows.  Pure host arithmetic. */
int32_t achip_partition_blocks(const int64_t* weight, int32_t nBlocks, int32_t nParts, int32_t* starts);

#ifdef __cplusplus
}
#endif
#endif /* AIRCOMPRESSOR_HIP_H */
