/*
 * oracle.h -- CPU restatement of aircompressor's *Java* block codecs.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product (libaircompressor_hip.so, aircompressor_amd/).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it,
 * and only as the checker / the CPU baseline.
 *
 * Each function follows one reference routine (M/ = src/main/java/io/airlift/
 * compress/v3/ of airlift/aircompressor 3.8-SNAPSHOT), cited at its definition.
 * Pinning: see oracle/README.md (golden vectors of the reference's own tests).
 *
 * Result convention: >= 0 bytes written, < 0 an ACHIP status from
 * include/aircompressor_hip.h (class + 16*detail, negated); *err_off receives
 * the offset the Java MalformedInputException would carry.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* LZ4 -- M/lz4/Lz4RawCompressor.java, M/lz4/Lz4RawDecompressor.java */
int64_t orc_lz4_max_compressed_length(int64_t n);
int64_t orc_lz4_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap);
int64_t orc_lz4_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off);

/* Snappy -- M/snappy/SnappyRawCompressor.java, M/snappy/SnappyRawDecompressor.java */
int64_t orc_snappy_max_compressed_length(int64_t n);
int64_t orc_snappy_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap);
int64_t orc_snappy_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off);
int64_t orc_snappy_uncompressed_length(const uint8_t* in, int64_t in_len, int64_t* err_off);

/* Zstd -- M/zstd/ZstdFrameDecompressor.java (+Huffman, FseTableReader, FiniteStateEntropy, BitInputStream, XxHash64) */
int64_t orc_zstd_max_compressed_length(int64_t n);
int64_t orc_zstd_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off);
int64_t orc_zstd_decompressed_size(const uint8_t* in, int64_t in_len, int64_t* err_off);
/* Zstd level-3 encoder -- M/zstd/ZstdFrameCompressor.java and friends */
int64_t orc_zstd_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap);
/* ZstdOutputStream (M/zstd/ZstdOutputStream.java:30-221): what one write(buffer, 0, n) + close() put on the sink */
int64_t orc_zstd_stream_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap);
int64_t orc_zstd_stream_max_compressed_length(int64_t n);

/* test hooks for the frame-header KATs of T/zstd/TestCompressor.java:52-98 */
int32_t orc_zstd_write_frame_header(uint8_t* out14, int32_t inputSize, int32_t windowSize);
int32_t orc_zstd_read_frame_header(const uint8_t* in, int64_t in_len, int64_t* out4);

/* XXH64 -- M/zstd/XxHash64.java:182-291 */
uint64_t orc_xxh64(const uint8_t* in, int64_t len, uint64_t seed);

/* LZ4 frame container -- M/lz4/Lz4FrameCompression.java:70-343 over the block codec above */
int64_t orc_lz4frame_max_compressed_length(int64_t n);
int64_t orc_lz4frame_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap);
int64_t orc_lz4frame_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off);

/* x-snappy-framed streams -- M/snappy/SnappyFramedOutputStream.java, SnappyFramedInputStream.java, Crc32C.java (snappy_framed.c) */
int64_t orc_snappyframed_max_compressed_length(int64_t n);
int64_t orc_snappyframed_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap);
int64_t orc_snappyframed_decompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int64_t* err_off);
uint32_t orc_crc32c(const uint8_t* in, int64_t len);
uint32_t orc_masked_crc32c(const uint8_t* in, int64_t len);

/* Hadoop LZ4 / Snappy block streams -- M/lz4/Lz4Hadoop{Input,Output}Stream.java, M/snappy/SnappyHadoop{Input,Output}Stream.java
 * (hadoop_streams.c); codec 0 = LZ4, 1 = Snappy; bufferSize = the streams' buffer size (262144 unless configured) */
int64_t orc_hadoop_max_compressed_length(int32_t codec, int64_t n, int32_t bufferSize);
int64_t orc_hadoop_compress(int32_t codec, const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int32_t bufferSize);
int64_t orc_hadoop_decompress(int32_t codec, const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap, int32_t bufferSize, int64_t* err_off);

/* XXH32 -- M/xxhash/XxHash32JavaHasher.java:68-110,343-366 (public xxhash package; LZ4 frame checksums) */
uint32_t orc_xxh32(const uint8_t* in, int64_t len, uint32_t seed);

/* Synthetic data: the reference's test generator -- T/snappy/RandomGenerator.java:25-74 on java.util.Random(301) */
void orc_random_generator(double compression_ratio, uint8_t* out, int64_t len);

/* batch drivers used by bench.py's cpu_baseline leg (plain loops, one thread) */
void orc_zstd_enc_thread_free(void);
void orc_zstd_dec_thread_free(void);
/* cpu_baseline timing driver (misc.c): T pthreads on disjoint block ranges; returns plaintext bytes / second */
double orc_bench(int32_t op, const uint8_t* src_base, const int64_t* src_off, const int32_t* src_len, uint8_t* dst_base, const int64_t* dst_off,
                 const int32_t* dst_cap, int32_t n_blocks, int32_t threads, double seconds, double* passes, int64_t* failures);
int64_t orc_batch(int32_t op, const uint8_t* src_base, const int64_t* src_off, const int32_t* src_len,
                  uint8_t* dst_base, const int64_t* dst_off, const int32_t* dst_cap,
                  int32_t* out_len, int32_t* status, int64_t* err_off, int32_t n_blocks);

#ifdef __cplusplus
}
#endif
#endif
