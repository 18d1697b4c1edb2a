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
package io.airlift.compress.v3.hadoop;

import io.airlift.compress.v3.MalformedInputException;
import io.airlift.compress.v3.hip.HipNative;

import java.io.EOFException;
import java.io.IOException;
import java.lang.foreign.MemorySegment;
import java.util.Arrays;

import static java.util.Objects.requireNonNull;

/**
 * Hadoop LZ4 / Snappy block streams on the GPU, whole buffers at a time: what
 * {@code HadoopStreams.createOutputStream(out); write(data); close()} leaves in {@code out}
 * ({@code Lz4HadoopOutputStream} / {@code SnappyHadoopOutputStream}: per chunk of {@code bufferSize - overhead} bytes a big-endian
 * plaintext length, a big-endian compressed length and the codec's block) and what reading
 * {@code HadoopStreams.createInputStream(in)} to its end returns ({@code Lz4HadoopInputStream} / {@code SnappyHadoopInputStream}:
 * blocks of several chunks, empty blocks, their IOException / EOFException for truncated streams), with the HIP block codecs underneath
 * ({@code achip_lz4hadoop_*} / {@code achip_snappyhadoop_*}, {@code OP_LZ4HADOOP_*} / {@code OP_SNAPPYHADOOP_*} for batches of streams).
 * A {@code HadoopStreams} implementation that buffers whole files (e.g. an ORC / SequenceFile reader that holds a stripe in memory)
 * can hand them to this class instead of pulling them through the stream classes block by block.
 *
 * NOT COMPILED IN THIS REPOSITORY (no JDK in the build image): see INTEGRATION.md.
 */
public final class HadoopBlockStreamsHip
{
    public enum Codec
    {
        LZ4(0, HipNative.OP_LZ4HADOOP_COMPRESS, HipNative.OP_LZ4HADOOP_DECOMPRESS),
        SNAPPY(1, HipNative.OP_SNAPPYHADOOP_COMPRESS, HipNative.OP_SNAPPYHADOOP_DECOMPRESS);

        final int id;
        final int compressOp;
        final int decompressOp;

        Codec(int id, int compressOp, int decompressOp)
        {
            this.id = id;
            this.compressOp = compressOp;
            this.decompressOp = decompressOp;
        }
    }

    /** the stream classes' default (Lz4HadoopStreams.java:30, SnappyHadoopStreams.java:30) */
    public static final int DEFAULT_BUFFER_SIZE = 256 * 1024;

    private final HipNative.Context context;
    private final Codec codec;
    private final int bufferSize;

    public HadoopBlockStreamsHip(Codec codec, int device, int bufferSize)
    {
        HipNative.verifyEnabled();
        this.codec = requireNonNull(codec, "codec is null");
        this.bufferSize = bufferSize;
        this.context = new HipNative.Context(device);
        this.context.setOption("hadoop.buffer_size", bufferSize);
    }

    public HadoopBlockStreamsHip(Codec codec)
    {
        this(codec, 0, DEFAULT_BUFFER_SIZE);
    }

    public static boolean isEnabled()
    {
        return HipNative.isEnabled();
    }

    public int maxCompressedLength(int uncompressedSize)
    {
        return HipNative.hadoopMaxCompressedLength(codec.id, uncompressedSize, bufferSize);
    }

    /** {@code createOutputStream(out); write(data); close()} */
    public byte[] compress(byte[] data)
    {
        requireNonNull(data, "data is null");
        byte[] output = new byte[maxCompressedLength(data.length)];
        int written = context.singleBlock(codec.compressOp, MemorySegment.ofArray(data), data.length, MemorySegment.ofArray(output), output.length);
        return Arrays.copyOf(output, written);
    }

    /**
     * {@code createInputStream(in)} read to its end into at most {@code maxUncompressedLength} bytes.  The stream classes' own failures
     * come back as the exceptions they throw: EOFException("encountered EOF while reading block data"), IOException("Stream is
     * truncated" / "Chunk uncompressed size is greater than block size" / ...); a destination that cannot hold the stream is an
     * IllegalArgumentException; everything else is the block codec's MalformedInputException.
     */
    public byte[] decompress(byte[] stream, int maxUncompressedLength)
            throws IOException
    {
        requireNonNull(stream, "stream is null");
        byte[] output = new byte[maxUncompressedLength];
        try {
            int written = context.singleBlock(codec.decompressOp, MemorySegment.ofArray(stream), stream.length, MemorySegment.ofArray(output), output.length);
            return Arrays.copyOf(output, written);
        }
        catch (MalformedInputException e) {
            String message = e.getMessage();
            if (message.startsWith(HipNative.detailMessage(105))) {  // ACHIP_D_HDP_EOF_BLOCK_DATA
                throw new EOFException(HipNative.detailMessage(105));
            }
            for (int detail : new int[] {104, 106, 107, 109}) {  // ACHIP_D_HDP_TRUNCATED_INT, _CHUNK_EXCEEDS_BLOCK, _LENGTH_MISMATCH, _NEGATIVE_LENGTH
                if (message.startsWith(HipNative.detailMessage(detail))) {
                    throw new IOException(HipNative.detailMessage(detail));
                }
            }
            throw e;
        }
    }
}
