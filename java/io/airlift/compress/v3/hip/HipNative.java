/*
 * Licensed under the Apache License, Version 2.0 (the "License");
 * you may not use this file except in compliance with the License.
 * You may obtain a copy of the License at
 *
 *     http://www.apache.org/licenses/LICENSE-2.0
 *
 * Unless required by applicable law or agreed to in writing, software
 * distributed under the License is distributed on an "AS IS" BASIS,
 * WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND, either express or implied.
 * See the License for the specific language governing permissions and
 * limitations under the License.
 */
package io.airlift.compress.v3.hip;

import io.airlift.compress.v3.MalformedInputException;
import io.airlift.compress.v3.internal.NativeLoader.Symbols;
import io.airlift.compress.v3.internal.NativeSignature;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.lang.invoke.MethodHandle;
import java.lang.ref.Cleaner;
import java.util.Optional;

import static io.airlift.compress.v3.internal.NativeLoader.loadSymbols;
import static java.lang.foreign.ValueLayout.JAVA_LONG;
import static java.lang.invoke.MethodHandles.lookup;

/**
 * FFM binding of {@code libaircompressor_hip.so} (C ABI: {@code include/aircompressor_hip.h}).
 * <p>
 * Built exactly like {@code Lz4Native}/{@code SnappyNative}/{@code ZstdNative}: a record of
 * {@link NativeSignature}-annotated method handles resolved by {@code NativeLoader.loadSymbols},
 * which extracts {@code /aircompressor/linux-amd64/libaircompressor_hip.so} from the class path.
 * When the library or a HIP device is missing every handle throws {@link LinkageError} and
 * {@link #isEnabled()} is false, so callers can fall back to the Java codecs the same way the
 * {@code create()} factories do for the existing native codecs.
 */
public final class HipNative
{
    private HipNative() {}

    // status = -(class + 16 * detail), see aircompressor_hip.h
    public static final int CLASS_MALFORMED = 1;
    public static final int CLASS_OUTPUT_TOO_SMALL = 2;
    public static final int CLASS_INVALID_ARGUMENT = 3;
    public static final int CLASS_DEVICE = 4;
    public static final int DETAIL_LZ4_EMPTY_OUTPUT = 7;

    public static final int OP_LZ4_DECOMPRESS = 0;
    public static final int OP_LZ4_COMPRESS = 1;
    public static final int OP_SNAPPY_DECOMPRESS = 2;
    public static final int OP_SNAPPY_COMPRESS = 3;
    public static final int OP_ZSTD_DECOMPRESS = 4;
    public static final int OP_ZSTD_COMPRESS = 5;
    public static final int OP_LZ4FRAME_DECOMPRESS = 6;  // achip_lz4frame_decompress (SURVEY 8f row 1)
    public static final int OP_LZ4FRAME_COMPRESS = 7;    // achip_lz4frame_compress
    public static final int OP_SNAPPYFRAMED_DECOMPRESS = 8;  // achip_snappyframed_decompress (SURVEY 8f row 2)
    public static final int OP_SNAPPYFRAMED_COMPRESS = 9;    // achip_snappyframed_compress
    public static final int OP_LZ4HADOOP_DECOMPRESS = 10;    // achip_lz4hadoop_decompress (SURVEY 8f row 2: Hadoop block streams)
    public static final int OP_LZ4HADOOP_COMPRESS = 11;      // achip_lz4hadoop_compress
    public static final int OP_SNAPPYHADOOP_DECOMPRESS = 12; // achip_snappyhadoop_decompress
    public static final int OP_SNAPPYHADOOP_COMPRESS = 13;   // achip_snappyhadoop_compress
    public static final int OP_ZSTDSTREAM_COMPRESS = 14;     // achip_zstdstream_compress (SURVEY 8f row 3: what a ZstdOutputStream puts on its sink)

    private record MethodHandles(
            @NativeSignature(name = "achip_device_count", returnType = int.class, argumentTypes = {})
            MethodHandle deviceCount,
            @NativeSignature(name = "achip_detail_message", returnType = MemorySegment.class, argumentTypes = int.class)
            MethodHandle detailMessage,
            @NativeSignature(name = "achip_last_error", returnType = MemorySegment.class, argumentTypes = {})
            MethodHandle lastError,
            @NativeSignature(name = "achip_lz4_max_compressed_length", returnType = int.class, argumentTypes = int.class)
            MethodHandle lz4MaxCompressedLength,
            @NativeSignature(name = "achip_snappy_max_compressed_length", returnType = int.class, argumentTypes = int.class)
            MethodHandle snappyMaxCompressedLength,
            @NativeSignature(name = "achip_zstd_max_compressed_length", returnType = int.class, argumentTypes = int.class)
            MethodHandle zstdMaxCompressedLength,
            @NativeSignature(name = "achip_snappy_uncompressed_length", returnType = long.class, argumentTypes = {MemorySegment.class, long.class, MemorySegment.class})
            MethodHandle snappyUncompressedLength,
            @NativeSignature(name = "achip_zstd_decompressed_size", returnType = long.class, argumentTypes = {MemorySegment.class, long.class, MemorySegment.class})
            MethodHandle zstdDecompressedSize,
            @NativeSignature(name = "achip_zstd_decompress_bound", returnType = long.class, argumentTypes = {MemorySegment.class, long.class, MemorySegment.class})
            MethodHandle zstdDecompressBound,
            @NativeSignature(name = "achip_ctx_create", returnType = MemorySegment.class, argumentTypes = int.class)
            MethodHandle ctxCreate,
            @NativeSignature(name = "achip_ctx_destroy", returnType = void.class, argumentTypes = MemorySegment.class)
            MethodHandle ctxDestroy,
            @NativeSignature(name = "achip_ctx_synchronize", returnType = int.class, argumentTypes = MemorySegment.class)
            MethodHandle ctxSynchronize,
            @NativeSignature(name = "achip_device_alloc", returnType = MemorySegment.class, argumentTypes = {MemorySegment.class, long.class})
            MethodHandle deviceAlloc,
            @NativeSignature(name = "achip_device_free", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class})
            MethodHandle deviceFree,
            @NativeSignature(name = "achip_host_alloc_pinned", returnType = MemorySegment.class, argumentTypes = long.class)
            MethodHandle hostAllocPinned,
            @NativeSignature(name = "achip_host_free_pinned", returnType = int.class, argumentTypes = MemorySegment.class)
            MethodHandle hostFreePinned,
            @NativeSignature(name = "achip_memcpy_h2d", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, long.class})
            MethodHandle memcpyHostToDevice,
            @NativeSignature(name = "achip_memcpy_d2h", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, long.class})
            MethodHandle memcpyDeviceToHost,
            // single block, host pointers: (ctx, src, dst, srcLen, dstCap, errOffset*) -- argument order of LZ4_compress_fast / LZ4_decompress_safe
            @NativeSignature(name = "achip_lz4_compress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle lz4Compress,
            @NativeSignature(name = "achip_lz4_decompress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle lz4Decompress,
            @NativeSignature(name = "achip_snappy_compress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle snappyCompress,
            @NativeSignature(name = "achip_snappy_decompress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle snappyDecompress,
            @NativeSignature(name = "achip_zstd_compress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle zstdCompress,
            @NativeSignature(name = "achip_zstd_decompress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle zstdDecompress,
            @NativeSignature(name = "achip_lz4frame_compress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle lz4FrameCompress,
            @NativeSignature(name = "achip_lz4frame_decompress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle lz4FrameDecompress,
            @NativeSignature(name = "achip_snappyframed_compress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle snappyFramedCompress,
            @NativeSignature(name = "achip_snappyframed_decompress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle snappyFramedDecompress,
            @NativeSignature(name = "achip_lz4frame_max_compressed_length", returnType = int.class, argumentTypes = int.class)
            MethodHandle lz4FrameMaxCompressedLength,
            @NativeSignature(name = "achip_snappyframed_max_compressed_length", returnType = int.class, argumentTypes = int.class)
            MethodHandle snappyFramedMaxCompressedLength,
            // Hadoop block streams (SURVEY 8f row 2, second half)
            @NativeSignature(name = "achip_lz4hadoop_compress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle lz4HadoopCompress,
            @NativeSignature(name = "achip_lz4hadoop_decompress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle lz4HadoopDecompress,
            @NativeSignature(name = "achip_snappyhadoop_compress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle snappyHadoopCompress,
            @NativeSignature(name = "achip_snappyhadoop_decompress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle snappyHadoopDecompress,
            @NativeSignature(name = "achip_hadoop_max_compressed_length", returnType = int.class, argumentTypes = {int.class, int.class, int.class})
            MethodHandle hadoopMaxCompressedLength,
            // Zstd streams (SURVEY 8f row 3)
            @NativeSignature(name = "achip_zstdstream_compress", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle zstdStreamCompress,
            @NativeSignature(name = "achip_zstdstream_max_compressed_length", returnType = int.class, argumentTypes = int.class)
            MethodHandle zstdStreamMaxCompressedLength,
            @NativeSignature(name = "achip_zstdstream_compress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle zstdStreamCompressBatch,
            @NativeSignature(name = "achip_ctx_set_option", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, long.class})
            MethodHandle ctxSetOption,
            // the public xxhash package on the GPU (SURVEY 8f row 4): (ctx, data, length, seed, out*) and (ctx, base, offsets*, lengths*, seed, hashes*, count)
            @NativeSignature(name = "achip_xxhash64", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, long.class, long.class, MemorySegment.class})
            MethodHandle xxhash64,
            @NativeSignature(name = "achip_xxhash32", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, long.class, int.class, MemorySegment.class})
            MethodHandle xxhash32,
            @NativeSignature(name = "achip_xxhash64_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, long.class, MemorySegment.class, int.class})
            MethodHandle xxhash64Batch,
            @NativeSignature(name = "achip_xxhash32_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, MemorySegment.class, int.class})
            MethodHandle xxhash32Batch,
            // batched, device-resident: (op, ctx, srcBase, srcOff*, srcLen*, dstBase, dstOff*, dstCap*, outLen*, status*, errOffset*, nBlocks)
            @NativeSignature(name = "achip_batch_host", returnType = int.class, argumentTypes = {int.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle batchHost,
            @NativeSignature(name = "achip_lz4_decompress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle lz4DecompressBatch,
            @NativeSignature(name = "achip_lz4_compress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle lz4CompressBatch,
            @NativeSignature(name = "achip_snappy_decompress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle snappyDecompressBatch,
            @NativeSignature(name = "achip_snappy_compress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle snappyCompressBatch,
            @NativeSignature(name = "achip_zstd_decompress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle zstdDecompressBatch,
            // containers, batched and device-resident: one whole frame / stream per item (same argument list)
            @NativeSignature(name = "achip_lz4frame_decompress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle lz4FrameDecompressBatch,
            @NativeSignature(name = "achip_lz4frame_compress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle lz4FrameCompressBatch,
            @NativeSignature(name = "achip_snappyframed_decompress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle snappyFramedDecompressBatch,
            @NativeSignature(name = "achip_snappyframed_compress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle snappyFramedCompressBatch,
            @NativeSignature(name = "achip_lz4hadoop_decompress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle lz4HadoopDecompressBatch,
            @NativeSignature(name = "achip_lz4hadoop_compress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle lz4HadoopCompressBatch,
            @NativeSignature(name = "achip_snappyhadoop_decompress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle snappyHadoopDecompressBatch,
            @NativeSignature(name = "achip_snappyhadoop_compress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle snappyHadoopCompressBatch,
            @NativeSignature(name = "achip_zstd_compress_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle zstdCompressBatch,
            // one process, several contexts: (ctxs**, nCtx, op, ops*, srcBase, srcOff*, srcLen*, dstBase, dstOff*, dstCap*, outLen*, status*, errOffset*, nBlocks, sliceStarts*)
            @NativeSignature(name = "achip_multi_batch_host", returnType = int.class, argumentTypes = {MemorySegment.class, int.class, int.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    int.class, MemorySegment.class})
            MethodHandle multiBatchHost,
            // mixed batches (BASELINE configs[4]: item i through codecOps[i]): (ctx, codecOps* [HOST], srcBase, srcOff*, srcLen*, dstBase, dstOff*, dstCap*, outLen*, status*, errOffset*, nBlocks);
            // achip_mixed_batch: everything but codecOps device-accessible, asynchronous on the context's stream; achip_mixed_batch_host: host memory, synchronous
            @NativeSignature(name = "achip_mixed_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle mixedBatch,
            @NativeSignature(name = "achip_mixed_batch_host", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class,
                    MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class})
            MethodHandle mixedBatchHost,
            // Zstd streams a step at a time (SURVEY 8f row 3): begin(ctx) -> state; feed(ctx, state, src, srcLen, dst, dstCap, consumed*, produced*[, errOffset*]); ...
            @NativeSignature(name = "achip_zstdstream_decompress_begin", returnType = MemorySegment.class, argumentTypes = MemorySegment.class)
            MethodHandle zstdStreamDecompressBegin,
            @NativeSignature(name = "achip_zstdstream_decompress_feed", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, long.class,
                    MemorySegment.class, long.class, MemorySegment.class, MemorySegment.class, MemorySegment.class})
            MethodHandle zstdStreamDecompressFeed,
            @NativeSignature(name = "achip_zstdstream_decompress_at_stopping_point", returnType = int.class, argumentTypes = MemorySegment.class)
            MethodHandle zstdStreamDecompressAtStoppingPoint,
            @NativeSignature(name = "achip_zstdstream_decompress_end", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class})
            MethodHandle zstdStreamDecompressEnd,
            @NativeSignature(name = "achip_zstdstream_compress_begin", returnType = MemorySegment.class, argumentTypes = MemorySegment.class)
            MethodHandle zstdStreamCompressBegin,
            @NativeSignature(name = "achip_zstdstream_compress_feed", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, long.class,
                    MemorySegment.class, long.class, MemorySegment.class, MemorySegment.class})
            MethodHandle zstdStreamCompressFeed,
            @NativeSignature(name = "achip_zstdstream_compress_finish", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, long.class,
                    MemorySegment.class})
            MethodHandle zstdStreamCompressFinish,
            @NativeSignature(name = "achip_zstdstream_compress_end", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class})
            MethodHandle zstdStreamCompressEnd,
            // the helpers of the boundary (round 6: every symbol the header declares is bound): status words taken apart by the library, its version string,
            // a context's device / stream / statistics, a device memset, events for timing on the context's stream, the multi-GPU split
            @NativeSignature(name = "achip_status_class", returnType = int.class, argumentTypes = int.class)
            MethodHandle statusClassNative,
            @NativeSignature(name = "achip_status_detail", returnType = int.class, argumentTypes = int.class)
            MethodHandle statusDetailNative,
            @NativeSignature(name = "achip_version", returnType = MemorySegment.class, argumentTypes = {})
            MethodHandle version,
            @NativeSignature(name = "achip_ctx_device", returnType = int.class, argumentTypes = MemorySegment.class)
            MethodHandle ctxDevice,
            @NativeSignature(name = "achip_ctx_stream", returnType = MemorySegment.class, argumentTypes = MemorySegment.class)
            MethodHandle ctxStream,
            @NativeSignature(name = "achip_ctx_get_stat", returnType = long.class, argumentTypes = {MemorySegment.class, MemorySegment.class})
            MethodHandle ctxGetStat,
            @NativeSignature(name = "achip_memset_d", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, int.class, long.class})
            MethodHandle memsetDevice,
            @NativeSignature(name = "achip_event_create", returnType = MemorySegment.class, argumentTypes = {})
            MethodHandle eventCreate,
            @NativeSignature(name = "achip_event_destroy", returnType = int.class, argumentTypes = MemorySegment.class)
            MethodHandle eventDestroy,
            @NativeSignature(name = "achip_event_record", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class})
            MethodHandle eventRecord,
            @NativeSignature(name = "achip_event_elapsed_ms", returnType = float.class, argumentTypes = {MemorySegment.class, MemorySegment.class})
            MethodHandle eventElapsedMs,
            @NativeSignature(name = "achip_partition_blocks", returnType = int.class, argumentTypes = {MemorySegment.class, int.class, int.class, MemorySegment.class})
            MethodHandle partitionBlocks) {}

    private static final Optional<LinkageError> LINKAGE_ERROR;
    private static final MethodHandles HANDLES;
    private static final int DEVICE_COUNT;
    private static final Cleaner CLEANER = Cleaner.create();

    static {
        Symbols<MethodHandles> symbols;
        if (System.getProperty("io.airlift.compress.v3.disable-hip") != null) {
            // mirrors io.airlift.compress.v3.disable-native (NativeLoader.java:158)
            symbols = new Symbols<>(Optional.of(new LinkageError("HIP backend is disabled")), null);
        }
        else {
            symbols = loadSymbols("aircompressor_hip", MethodHandles.class, lookup());
        }
        LINKAGE_ERROR = symbols.linkageError();
        HANDLES = symbols.symbols();
        int devices = 0;
        if (LINKAGE_ERROR.isEmpty()) {
            try {
                devices = (int) HANDLES.deviceCount().invokeExact();
            }
            catch (Throwable e) {
                throw new ExceptionInInitializerError(e);
            }
        }
        DEVICE_COUNT = devices;
    }

    /** true when the library loaded AND at least one HIP device is usable. */
    public static boolean isEnabled()
    {
        return LINKAGE_ERROR.isEmpty() && DEVICE_COUNT > 0;
    }

    public static int deviceCount()
    {
        return DEVICE_COUNT;
    }

    public static void verifyEnabled()
    {
        if (LINKAGE_ERROR.isPresent()) {
            throw new IllegalStateException("HIP native library is not enabled", LINKAGE_ERROR.get());
        }
        if (DEVICE_COUNT <= 0) {
            throw new IllegalStateException("No HIP device available");
        }
    }

    // ---- status translation -------------------------------------------------------------------

    static int statusClass(int status)
    {
        return status < 0 ? ((-status) & 15) : 0;
    }

    static int statusDetail(int status)
    {
        return status < 0 ? ((-status) >> 4) : 0;
    }

    public static String detailMessage(int detail)
    {
        try {
            MemorySegment text = (MemorySegment) HANDLES.detailMessage().invokeExact(detail);
            return text.reinterpret(256).getString(0);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    /**
     * Throws what the Java codec would throw for this status: MalformedInputException(offset, reason)
     * for corrupt input, IllegalArgumentException for buffer sizing / arguments.
     */
    /**
     * One call, several contexts (normally one per device): the library cuts the batch into contiguous slices balanced by bytes and runs
     * every slice on its context in a host thread of its own ({@code achip_multi_batch_host}).  {@code ops} is {@link MemorySegment#NULL}
     * for a homogeneous batch of {@code op}, otherwise one OP_* per item.
     */
    public static void multiBatchHost(Context[] contexts, int op, MemorySegment ops, MemorySegment srcBase, MemorySegment srcOff, MemorySegment srcLen,
            MemorySegment dstBase, MemorySegment dstOff, MemorySegment dstCap, MemorySegment outLen, MemorySegment status, MemorySegment errOffset, int blocks,
            MemorySegment sliceStarts)
    {
        verifyEnabled();
        int result;
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment handles = arena.allocate(java.lang.foreign.ValueLayout.ADDRESS, contexts.length);
            for (int i = 0; i < contexts.length; i++) {
                handles.setAtIndex(java.lang.foreign.ValueLayout.ADDRESS, i, contexts[i].handle());
            }
            result = (int) HANDLES.multiBatchHost().invokeExact(handles, contexts.length, op, ops, srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset,
                    blocks, sliceStarts);
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
        if (result < 0) {
            throw toException(result, 0);
        }
    }

    /** achip_version: the library's version string */
    public static String version()
    {
        verifyEnabled();
        try {
            MemorySegment text = (MemorySegment) HANDLES.version().invokeExact();
            return text.reinterpret(256).getString(0);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    /**
     * achip_partition_blocks: the contiguous split of a batch over {@code parts} GPUs, balanced by {@code weight} (bytes moved per block) -- the rule
     * {@code achip_multi_batch_host} applies inside the library, so that a caller that launches the slices itself cuts where the library would.
     * Returns {@code parts + 1} block indices.  Pure host arithmetic: no device is needed.
     */
    public static int[] partitionBlocks(long[] weight, int parts)
    {
        if (LINKAGE_ERROR.isPresent()) {
            throw new IllegalStateException("HIP native library is not enabled", LINKAGE_ERROR.get());
        }
        int result;
        int[] starts = new int[parts + 1];
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment weights = arena.allocateFrom(java.lang.foreign.ValueLayout.JAVA_LONG, weight);
            MemorySegment out = arena.allocate(java.lang.foreign.ValueLayout.JAVA_INT, parts + 1);
            result = (int) HANDLES.partitionBlocks().invokeExact(weights, weight.length, parts, out);
            MemorySegment.copy(out, java.lang.foreign.ValueLayout.JAVA_INT, 0, starts, 0, parts + 1);
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
        if (result < 0) {
            throw toException(result, 0);
        }
        return starts;
    }

    /** achip_status_class / achip_status_detail: a status word taken apart by the library itself (the Java arithmetic above must agree: asserted by the callers' tests) */
    public static int nativeStatusClass(int status)
    {
        try {
            return (int) HANDLES.statusClassNative().invokeExact(status);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    public static int nativeStatusDetail(int status)
    {
        try {
            return (int) HANDLES.statusDetailNative().invokeExact(status);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    /** An event on a context's stream (achip_event_*): two of them time what was enqueued between their {@link Context#record} calls. */
    public static final class Event
            implements AutoCloseable
    {
        private final MemorySegment handle;
        private final java.util.concurrent.atomic.AtomicBoolean destroyed = new java.util.concurrent.atomic.AtomicBoolean();

        public Event()
        {
            verifyEnabled();
            try {
                handle = (MemorySegment) HANDLES.eventCreate().invokeExact();
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            if (handle.equals(MemorySegment.NULL)) {
                throw new IllegalStateException("achip_event_create failed: " + lastError());
            }
        }

        MemorySegment handle()
        {
            if (destroyed.get()) {
                throw new IllegalStateException("HipNative.Event is closed");
            }
            return handle;
        }

        /** milliseconds between this event and {@code stop}, both recorded and reached (achip_event_elapsed_ms: negative on failure) */
        public float elapsedMillis(Event stop)
        {
            try {
                return (float) HANDLES.eventElapsedMs().invokeExact(handle(), stop.handle());
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }

        @Override
        public void close()
        {
            if (destroyed.compareAndSet(false, true)) {
                try {
                    int ignored = (int) HANDLES.eventDestroy().invokeExact(handle);
                }
                catch (Throwable e) {
                    throw new AssertionError("should not reach here", e);
                }
            }
        }
    }

    public static RuntimeException toException(int status, long errorOffset)
    {
        String reason = detailMessage(statusDetail(status));
        return switch (statusClass(status)) {
            case CLASS_MALFORMED -> new MalformedInputException(errorOffset, reason);
            case CLASS_DEVICE -> new IllegalStateException(reason + ": " + lastError());
            default -> new IllegalArgumentException(reason);
        };
    }

    static String lastError()
    {
        try {
            MemorySegment text = (MemorySegment) HANDLES.lastError().invokeExact();
            return text.reinterpret(512).getString(0);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    // ---- size helpers -------------------------------------------------------------------------

    public static int lz4MaxCompressedLength(int n)
    {
        try {
            return (int) HANDLES.lz4MaxCompressedLength().invokeExact(n);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    public static int snappyMaxCompressedLength(int n)
    {
        try {
            return (int) HANDLES.snappyMaxCompressedLength().invokeExact(n);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    public static int lz4FrameMaxCompressedLength(int n)
    {
        try {
            int result = (int) HANDLES.lz4FrameMaxCompressedLength().invokeExact(n);
            if (result < 0) {
                throw new IllegalArgumentException(n < 0 ? "uncompressedSize is negative: " + n : "Maximum compressed length exceeds Integer.MAX_VALUE for uncompressedSize: " + n);
            }
            return result;
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    public static int snappyFramedMaxCompressedLength(int n)
    {
        try {
            int result = (int) HANDLES.snappyFramedMaxCompressedLength().invokeExact(n);
            if (result < 0) {
                throw new IllegalArgumentException(n < 0 ? "uncompressedSize is negative: " + n : "Maximum compressed length exceeds Integer.MAX_VALUE for uncompressedSize: " + n);
            }
            return result;
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    /** bound of what a ZstdOutputStream writes for n bytes (achip_zstdstream_max_compressed_length) */
    public static int zstdStreamMaxCompressedLength(int n)
    {
        try {
            int result = (int) HANDLES.zstdStreamMaxCompressedLength().invokeExact(n);
            if (result < 0) {
                throw new IllegalArgumentException(n < 0 ? "uncompressedSize is negative: " + n : "Maximum compressed length exceeds Integer.MAX_VALUE for uncompressedSize: " + n);
            }
            return result;
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    /** bound the one-shot Hadoop stream writers ask of their destination: codec 0 = LZ4, 1 = Snappy (achip_hadoop_max_compressed_length) */
    public static int hadoopMaxCompressedLength(int codec, int n, int bufferSize)
    {
        try {
            int result = (int) HANDLES.hadoopMaxCompressedLength().invokeExact(codec, n, bufferSize);
            if (result < 0) {
                throw new IllegalArgumentException(n < 0 ? "uncompressedSize is negative: " + n : "Maximum compressed length exceeds Integer.MAX_VALUE for uncompressedSize: " + n);
            }
            return result;
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    // handles of the batched / one-shot hashers (io.airlift.compress.v3.xxhash.XxHashHip)
    public static MethodHandle xxhash64()
    {
        return HANDLES.xxhash64();
    }

    public static MethodHandle xxhash32()
    {
        return HANDLES.xxhash32();
    }

    public static MethodHandle xxhash64Batch()
    {
        return HANDLES.xxhash64Batch();
    }

    public static MethodHandle xxhash32Batch()
    {
        return HANDLES.xxhash32Batch();
    }

    /** throws what {@link #toException} builds when {@code status} is negative */
    public static void throwIfError(int status, long errorOffset)
    {
        if (status < 0) {
            throw toException(status, errorOffset);
        }
    }

    public static int zstdMaxCompressedLength(int n)
    {
        try {
            return (int) HANDLES.zstdMaxCompressedLength().invokeExact(n);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    public static long snappyUncompressedLength(MemorySegment compressed)
    {
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment errorOffset = arena.allocate(JAVA_LONG);
            long result = (long) HANDLES.snappyUncompressedLength().invokeExact(compressed, compressed.byteSize(), errorOffset);
            if (result < 0) {
                throw toException((int) result, errorOffset.get(JAVA_LONG, 0));
            }
            return result;
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    /** An upper bound of what all frames in {@code compressed} decode to, from their frame and block headers (frames need no content size). */
    public static long zstdDecompressBound(MemorySegment compressed)
    {
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment errorOffset = arena.allocate(JAVA_LONG);
            long result = (long) HANDLES.zstdDecompressBound().invokeExact(compressed, compressed.byteSize(), errorOffset);
            if (result < 0) {
                throw toException((int) result, errorOffset.get(JAVA_LONG, 0));
            }
            return result;
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    public static long zstdDecompressedSize(MemorySegment compressed)
    {
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment errorOffset = arena.allocate(JAVA_LONG);
            long result = (long) HANDLES.zstdDecompressedSize().invokeExact(compressed, compressed.byteSize(), errorOffset);
            if (result < -1) {
                throw toException((int) result, errorOffset.get(JAVA_LONG, 0));
            }
            return result;
        }
        catch (RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    // ---- context --------------------------------------------------------------------------------

    /**
     * One HIP stream + device scratch on one GPU.  Like a codec instance it is not thread-safe;
     * distinct contexts may be used from distinct threads.  Freed by a Cleaner.
     */
    public static final class Context
            implements AutoCloseable
    {
        private final MemorySegment handle;
        private final java.util.concurrent.atomic.AtomicBoolean destroyed = new java.util.concurrent.atomic.AtomicBoolean();
        private final MemorySegment errorOffset = Arena.ofAuto().allocate(JAVA_LONG);
        private final java.lang.ref.Cleaner.Cleanable cleanable;

        public Context(int device)
        {
            verifyEnabled();
            MemorySegment created;
            try {
                created = (MemorySegment) HANDLES.ctxCreate().invokeExact(device);
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            if (created.address() == 0) {
                throw new IllegalStateException("achip_ctx_create(" + device + ") failed: " + lastError());
            }
            handle = created;
            MethodHandle destroy = HANDLES.ctxDestroy();
            java.util.concurrent.atomic.AtomicBoolean destroyed = this.destroyed;  // (the cleaner's action must not capture `this`)
            this.cleanable = CLEANER.register(this, () -> {
                if (destroyed.compareAndSet(false, true)) {
                    try {
                        destroy.invokeExact(created);
                    }
                    catch (Throwable ignored) {
                    }
                }
            });
        }

        /**
         * Frees the native context (HIP stream, device scratch) now; the cleaner is only the backstop for contexts nobody closed.
         * Every entry point of a closed context throws {@link IllegalStateException}: the native pointer is never handed out again.
         */
        @Override
        public void close()
        {
            cleanable.clean();
        }

        public boolean isClosed()
        {
            return destroyed.get();
        }

        /** the native context; throws once {@link #close()} has run (a freed pointer must not reach a downcall) */
        MemorySegment handle()
        {
            if (destroyed.get()) {
                throw new IllegalStateException("HipNative.Context is closed");
            }
            return handle;
        }

        public void synchronize()
        {
            int status;
            try {
                status = (int) HANDLES.ctxSynchronize().invokeExact(handle());
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            if (status < 0) {
                throw toException(status, 0);
            }
        }

        /** achip_ctx_device: the device ordinal this context was created on */
        public int device()
        {
            try {
                return (int) HANDLES.ctxDevice().invokeExact(handle());
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }

        /** achip_ctx_stream: the hipStream_t the context's batch calls are enqueued on (for callers that order their own device work against it) */
        public MemorySegment stream()
        {
            try {
                return (MemorySegment) HANDLES.ctxStream().invokeExact(handle());
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }

        /** achip_ctx_get_stat: a counter of the context's last call (e.g. "zstd.decompress.fallback_items"); -1 for a name the library does not know */
        public long getStat(String name)
        {
            try (Arena arena = Arena.ofConfined()) {
                return (long) HANDLES.ctxGetStat().invokeExact(handle(), arena.allocateFrom(name));
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }

        /** achip_memset_d: fills device memory on the context's stream */
        public void memsetDevice(MemorySegment destination, int value, long bytes)
        {
            int result;
            try {
                result = (int) HANDLES.memsetDevice().invokeExact(handle(), destination, value, bytes);
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            throwIfError(result, 0);
        }

        /** achip_event_record: the event is reached when everything enqueued on this context's stream so far has run */
        public void record(Event event)
        {
            int result;
            try {
                result = (int) HANDLES.eventRecord().invokeExact(handle(), event.handle());
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            throwIfError(result, 0);
        }

        /** achip_ctx_set_option: a tuning / format option of this context (e.g. "hadoop.buffer_size") */
        public void setOption(String name, long value)
        {
            try (Arena arena = Arena.ofConfined()) {
                int result = (int) HANDLES.ctxSetOption().invokeExact(handle(), arena.allocateFrom(name), value);
                throwIfError(result, 0);
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }

        /** Single block through host memory: returns bytes written or throws the Java codec's exception. */
        public int singleBlock(int op, MemorySegment input, int inputLength, MemorySegment output, int outputLength)
        {
            int result;
            try {
                MethodHandle method = switch (op) {
                    case OP_LZ4_COMPRESS -> HANDLES.lz4Compress();
                    case OP_LZ4_DECOMPRESS -> HANDLES.lz4Decompress();
                    case OP_SNAPPY_COMPRESS -> HANDLES.snappyCompress();
                    case OP_SNAPPY_DECOMPRESS -> HANDLES.snappyDecompress();
                    case OP_ZSTD_COMPRESS -> HANDLES.zstdCompress();
                    case OP_ZSTD_DECOMPRESS -> HANDLES.zstdDecompress();
                    case OP_LZ4FRAME_COMPRESS -> HANDLES.lz4FrameCompress();
                    case OP_LZ4FRAME_DECOMPRESS -> HANDLES.lz4FrameDecompress();
                    case OP_SNAPPYFRAMED_COMPRESS -> HANDLES.snappyFramedCompress();
                    case OP_SNAPPYFRAMED_DECOMPRESS -> HANDLES.snappyFramedDecompress();
                    case OP_LZ4HADOOP_COMPRESS -> HANDLES.lz4HadoopCompress();
                    case OP_LZ4HADOOP_DECOMPRESS -> HANDLES.lz4HadoopDecompress();
                    case OP_SNAPPYHADOOP_COMPRESS -> HANDLES.snappyHadoopCompress();
                    case OP_ZSTDSTREAM_COMPRESS -> HANDLES.zstdStreamCompress();
                    case OP_SNAPPYHADOOP_DECOMPRESS -> HANDLES.snappyHadoopDecompress();
                    default -> throw new IllegalArgumentException("unknown op " + op);
                };
                result = (int) method.invokeExact(handle(), input, output, inputLength, outputLength, errorOffset);
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            if (result < 0) {
                if (op == OP_LZ4_DECOMPRESS && statusDetail(result) == DETAIL_LZ4_EMPTY_OUTPUT) {
                    return -1; // Lz4RawDecompressor.java:52-57 returns -1 here instead of throwing
                }
                throw toException(result, errorOffset.get(JAVA_LONG, 0));
            }
            return result;
        }

        /** Launches a device-resident batch (asynchronous on this context's stream). All segments are native (device or pinned). */
        public void launchBatch(int op, MemorySegment srcBase, MemorySegment srcOff, MemorySegment srcLen, MemorySegment dstBase, MemorySegment dstOff,
                MemorySegment dstCap, MemorySegment outLen, MemorySegment status, MemorySegment errOffset, int blocks)
        {
            int result;
            try {
                MethodHandle method = switch (op) {
                    case OP_LZ4_COMPRESS -> HANDLES.lz4CompressBatch();
                    case OP_LZ4_DECOMPRESS -> HANDLES.lz4DecompressBatch();
                    case OP_SNAPPY_COMPRESS -> HANDLES.snappyCompressBatch();
                    case OP_SNAPPY_DECOMPRESS -> HANDLES.snappyDecompressBatch();
                    case OP_ZSTD_COMPRESS -> HANDLES.zstdCompressBatch();
                    case OP_ZSTD_DECOMPRESS -> HANDLES.zstdDecompressBatch();
                    case OP_LZ4FRAME_COMPRESS -> HANDLES.lz4FrameCompressBatch();
                    case OP_LZ4FRAME_DECOMPRESS -> HANDLES.lz4FrameDecompressBatch();
                    case OP_SNAPPYFRAMED_COMPRESS -> HANDLES.snappyFramedCompressBatch();
                    case OP_SNAPPYFRAMED_DECOMPRESS -> HANDLES.snappyFramedDecompressBatch();
                    case OP_LZ4HADOOP_COMPRESS -> HANDLES.lz4HadoopCompressBatch();
                    case OP_LZ4HADOOP_DECOMPRESS -> HANDLES.lz4HadoopDecompressBatch();
                    case OP_SNAPPYHADOOP_COMPRESS -> HANDLES.snappyHadoopCompressBatch();
                    case OP_ZSTDSTREAM_COMPRESS -> HANDLES.zstdStreamCompressBatch();
                    case OP_SNAPPYHADOOP_DECOMPRESS -> HANDLES.snappyHadoopDecompressBatch();
                    default -> throw new IllegalArgumentException("unknown op " + op);
                };
                result = (int) method.invokeExact(handle(), srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, blocks);
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            if (result < 0) {
                throw toException(result, 0);
            }
        }

        /**
         * A batch whose items name their own codec and direction ({@code codecOps}: one OP_* int per item, HOST memory -- the caller's knowledge of its items;
         * everything else device-accessible as for {@link #launchBatch}): bucketed by op inside the library, the three codec families side by side, results in the
         * caller's item order.  Asynchronous on the context's stream.  (BASELINE configs[4]; an ORC / Parquet stripe with pages of three codecs.)
         */
        public void launchMixedBatch(MemorySegment codecOps, MemorySegment srcBase, MemorySegment srcOff, MemorySegment srcLen, MemorySegment dstBase, MemorySegment dstOff,
                MemorySegment dstCap, MemorySegment outLen, MemorySegment status, MemorySegment errOffset, int blocks)
        {
            int result;
            try {
                result = (int) HANDLES.mixedBatch().invokeExact(handle(), codecOps, srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, blocks);
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            if (result < 0) {
                throw toException(result, 0);
            }
        }

        /** The same with every array and both buffers in host memory: staged like {@link #batchHost}, synchronous. */
        public void mixedBatchHost(MemorySegment codecOps, MemorySegment srcBase, MemorySegment srcOff, MemorySegment srcLen, MemorySegment dstBase, MemorySegment dstOff,
                MemorySegment dstCap, MemorySegment outLen, MemorySegment status, MemorySegment errOffset, int blocks)
        {
            int result;
            try {
                result = (int) HANDLES.mixedBatchHost().invokeExact(handle(), codecOps, srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, blocks);
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            if (result < 0) {
                throw toException(result, 0);
            }
        }

        /** Host-memory batch: stages in, runs, stages out, synchronizes. Arrays are native or heap segments. */
        public void batchHost(int op, MemorySegment srcBase, MemorySegment srcOff, MemorySegment srcLen, MemorySegment dstBase, MemorySegment dstOff,
                MemorySegment dstCap, MemorySegment outLen, MemorySegment status, MemorySegment errOffset, int blocks)
        {
            int result;
            try {
                result = (int) HANDLES.batchHost().invokeExact(op, handle(), srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, blocks);
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
            if (result < 0) {
                throw toException(result, 0);
            }
        }

        /**
         * A Zstd stream decoded a step at a time in bounded memory ({@code achip_zstdstream_decompress_*}): what {@code ZstdInputStream} does over
         * {@code ZstdIncrementalFrameDecompressor}.  Not thread-safe; close it before its context.
         */
        public final class ZstdDecodeStream
                implements AutoCloseable
        {
            private MemorySegment state;
            private final Arena arena = Arena.ofConfined();
            private final MemorySegment counters = arena.allocate(JAVA_LONG, 3);  // consumed, produced, error offset

            ZstdDecodeStream()
            {
                try {
                    state = (MemorySegment) HANDLES.zstdStreamDecompressBegin().invokeExact(handle());
                }
                catch (Throwable e) {
                    throw new AssertionError("should not reach here", e);
                }
                if (state.address() == 0) {
                    throw new IllegalStateException("achip_zstdstream_decompress_begin failed: " + lastError());
                }
            }

            /** Takes input, delivers output; {@link #consumed()} / {@link #produced()} say how much.  Throws once the stream is damaged and every byte in front of the damage is out. */
            public void feed(MemorySegment input, long inputLength, MemorySegment output, long outputLength)
            {
                int result;
                try {
                    result = (int) HANDLES.zstdStreamDecompressFeed().invokeExact(handle(), state, input, inputLength, output, outputLength,
                            counters, counters.asSlice(8), counters.asSlice(16));
                }
                catch (Throwable e) {
                    throw new AssertionError("should not reach here", e);
                }
                if (result < 0) {
                    throw toException(result, counters.get(JAVA_LONG, 16));
                }
            }

            public long consumed()
            {
                return counters.get(JAVA_LONG, 0);
            }

            public long produced()
            {
                return counters.get(JAVA_LONG, 8);
            }

            /** Nothing is pending and the next byte would start a frame: where a stream may end ({@code ZstdIncrementalFrameDecompressor.isAtStoppingPoint}). */
            public boolean atStoppingPoint()
            {
                try {
                    return (int) HANDLES.zstdStreamDecompressAtStoppingPoint().invokeExact(state) != 0;
                }
                catch (Throwable e) {
                    throw new AssertionError("should not reach here", e);
                }
            }

            @Override
            public void close()
            {
                if (state != null) {
                    try {
                        int ignored = (int) HANDLES.zstdStreamDecompressEnd().invokeExact(handle(), state);
                    }
                    catch (Throwable e) {
                        throw new AssertionError("should not reach here", e);
                    }
                    state = null;
                    arena.close();
                }
            }
        }

        public ZstdDecodeStream openZstdDecodeStream()
        {
            return new ZstdDecodeStream();
        }

        /** A Zstd stream written a chunk at a time ({@code achip_zstdstream_compress_*}): {@code ZstdOutputStream}'s bytes in the 4 MiB it buffers. */
        public final class ZstdEncodeStream
                implements AutoCloseable
        {
            private MemorySegment state;
            private final Arena arena = Arena.ofConfined();
            private final MemorySegment counters = arena.allocate(JAVA_LONG, 2);  // consumed, produced

            ZstdEncodeStream()
            {
                try {
                    state = (MemorySegment) HANDLES.zstdStreamCompressBegin().invokeExact(handle());
                }
                catch (Throwable e) {
                    throw new AssertionError("should not reach here", e);
                }
                if (state.address() == 0) {
                    throw new IllegalStateException("achip_zstdstream_compress_begin failed: " + lastError());
                }
            }

            /** {@code write(input, 0, inputLength)} as far as {@code output} has room for the blocks flushed on the way. */
            public void feed(MemorySegment input, long inputLength, MemorySegment output, long outputLength)
            {
                int result;
                try {
                    result = (int) HANDLES.zstdStreamCompressFeed().invokeExact(handle(), state, input, inputLength, output, outputLength, counters, counters.asSlice(8));
                }
                catch (Throwable e) {
                    throw new AssertionError("should not reach here", e);
                }
                if (result < 0) {
                    throw toException(result, 0);
                }
            }

            /** {@code close()}: true once the stream's last byte has been delivered; false: {@code output} was too small for the rest, call again. */
            public boolean finish(MemorySegment output, long outputLength)
            {
                int result;
                try {
                    result = (int) HANDLES.zstdStreamCompressFinish().invokeExact(handle(), state, output, outputLength, counters.asSlice(8));
                }
                catch (Throwable e) {
                    throw new AssertionError("should not reach here", e);
                }
                if (result < 0) {
                    throw toException(result, 0);
                }
                return result == 1;
            }

            public long consumed()
            {
                return counters.get(JAVA_LONG, 0);
            }

            public long produced()
            {
                return counters.get(JAVA_LONG, 8);
            }

            @Override
            public void close()
            {
                if (state != null) {
                    try {
                        int ignored = (int) HANDLES.zstdStreamCompressEnd().invokeExact(handle(), state);
                    }
                    catch (Throwable e) {
                        throw new AssertionError("should not reach here", e);
                    }
                    state = null;
                    arena.close();
                }
            }
        }

        public ZstdEncodeStream openZstdEncodeStream()
        {
            return new ZstdEncodeStream();
        }

        public MemorySegment allocateDevice(long bytes)
        {
            try {
                MemorySegment segment = (MemorySegment) HANDLES.deviceAlloc().invokeExact(handle(), bytes);
                if (segment.address() == 0) {
                    throw new OutOfMemoryError("achip_device_alloc(" + bytes + ") failed: " + lastError());
                }
                return segment.reinterpret(bytes);
            }
            catch (Error | RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }

        public void freeDevice(MemorySegment segment)
        {
            try {
                int ignored = (int) HANDLES.deviceFree().invokeExact(handle(), segment);
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }

        public void copyToDevice(MemorySegment device, MemorySegment host, long bytes)
        {
            try {
                int status = (int) HANDLES.memcpyHostToDevice().invokeExact(handle(), device, host, bytes);
                if (status < 0) {
                    throw toException(status, 0);
                }
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }

        public void copyToHost(MemorySegment host, MemorySegment device, long bytes)
        {
            try {
                int status = (int) HANDLES.memcpyDeviceToHost().invokeExact(handle(), host, device, bytes);
                if (status < 0) {
                    throw toException(status, 0);
                }
            }
            catch (RuntimeException e) {
                throw e;
            }
            catch (Throwable e) {
                throw new AssertionError("should not reach here", e);
            }
        }
    }

    public static MemorySegment allocatePinned(long bytes)
    {
        try {
            MemorySegment segment = (MemorySegment) HANDLES.hostAllocPinned().invokeExact(bytes);
            if (segment.address() == 0) {
                throw new OutOfMemoryError("achip_host_alloc_pinned(" + bytes + ") failed");
            }
            return segment.reinterpret(bytes);
        }
        catch (Error | RuntimeException e) {
            throw e;
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }

    public static void freePinned(MemorySegment segment)
    {
        try {
            int ignored = (int) HANDLES.hostFreePinned().invokeExact(segment);
        }
        catch (Throwable e) {
            throw new AssertionError("should not reach here", e);
        }
    }
}
