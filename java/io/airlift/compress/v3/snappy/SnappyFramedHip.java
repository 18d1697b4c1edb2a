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
package io.airlift.compress.v3.snappy;

import io.airlift.compress.v3.MalformedInputException;
import io.airlift.compress.v3.hip.HipNative;

import java.io.IOException;
import java.lang.foreign.MemorySegment;
import java.util.Arrays;
import java.util.HashSet;
import java.util.Set;

import static java.util.Objects.requireNonNull;

/**
 * Whole-buffer x-snappy-framed codec on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}.
 * <p>
 * {@link #compress(byte[])} returns exactly what {@code new SnappyFramedOutputStream(new SnappyJavaCompressor(), out); write(data); close()}
 * leaves in {@code out} (stream header, then per 64 KiB block a chunk with the masked CRC-32C of its plaintext, compressed or -- when it
 * does not reach 0.85 -- raw); {@link #decompress(byte[], int)} returns what reading a {@link SnappyFramedInputStream} (checksums verified) to its
 * end returns, and fails where it fails: the stream-level errors come back as {@link IOException} with the stream class's message, a corrupt
 * chunk body as the block codec's {@code MalformedInputException}.
 * Bindings: {@code achip_snappyframed_compress}, {@code achip_snappyframed_decompress}, {@code achip_snappyframed_max_compressed_length}
 * (include/aircompressor_hip.h).  For many streams per call use {@link io.airlift.compress.v3.hip.HipBatchCodec} with
 * {@code OP_SNAPPYFRAMED_COMPRESS} / {@code OP_SNAPPYFRAMED_DECOMPRESS}.
 * <p>
 * Not thread-safe (owns one HIP stream).
 */
public final class SnappyFramedHip
{
    // ACHIP_D_SNF_EOF_STREAM_HEADER .. ACHIP_D_SNF_CHECKSUM: the errors SnappyFramedInputStream raises itself (IOException / EOFException)
    private static final Set<String> STREAM_LEVEL = new HashSet<>();

    static {
        for (int detail = 88; detail <= 95; detail++) {
            STREAM_LEVEL.add(HipNative.detailMessage(detail));
        }
    }

    private final HipNative.Context context;

    public SnappyFramedHip()
    {
        this(0);
    }

    public SnappyFramedHip(int device)
    {
        HipNative.verifyEnabled();
        this.context = new HipNative.Context(device);
    }

    public static boolean isEnabled()
    {
        return HipNative.isEnabled();
    }

    public static int maxCompressedLength(int uncompressedSize)
    {
        return HipNative.snappyFramedMaxCompressedLength(uncompressedSize);
    }

    public byte[] compress(byte[] data)
    {
        requireNonNull(data, "data is null");
        byte[] output = new byte[maxCompressedLength(data.length)];
        int written = context.singleBlock(HipNative.OP_SNAPPYFRAMED_COMPRESS, MemorySegment.ofArray(data), data.length, MemorySegment.ofArray(output), output.length);
        return Arrays.copyOf(output, written);
    }

    public byte[] decompress(byte[] stream, int maxUncompressedLength)
            throws IOException
    {
        requireNonNull(stream, "stream is null");
        byte[] output = new byte[maxUncompressedLength];
        try {
            int written = context.singleBlock(HipNative.OP_SNAPPYFRAMED_DECOMPRESS, MemorySegment.ofArray(stream), stream.length, MemorySegment.ofArray(output), output.length);
            return Arrays.copyOf(output, written);
        }
        catch (MalformedInputException e) {
            // the stream class throws IOException / EOFException for its own checks; everything else is the block codec's exception
            for (String reason : STREAM_LEVEL) {
                if (e.getMessage().startsWith(reason)) {
                    throw new IOException(reason);
                }
            }
            throw e;
        }
    }
}
