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
package io.airlift.compress.v3.lz4;

import io.airlift.compress.v3.Compressor;
import io.airlift.compress.v3.hip.HipNative;

import java.lang.foreign.MemorySegment;

import static java.lang.Math.toIntExact;
import static java.lang.String.format;
import static java.util.Objects.requireNonNull;

/**
 * LZ4 frame compressor running on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}.
 * Drop-in for {@code Lz4FrameJavaCompressor}: the frame {@code Lz4FrameCompression.compress} writes (independent 4 MiB
 * blocks, no checksums, blocks stored uncompressed when that is not smaller), byte for byte -- the blocks go through
 * the HIP block encoder, which restates the Java encoder's greedy parse step for step.
 * Binding: {@code achip_lz4frame_compress}, {@code achip_lz4frame_max_compressed_length} (include/aircompressor_hip.h).
 * <p>
 * This class is not thread-safe (it owns one HIP stream), like {@code Lz4JavaCompressor}.
 * For throughput use {@link io.airlift.compress.v3.hip.HipBatchCodec}: one call, many blocks, data resident on the device.
 */
public final class Lz4FrameHipCompressor
        implements Lz4FrameCompressor
{
    private final HipNative.Context context;

    public Lz4FrameHipCompressor()
    {
        this(0);
    }

    public Lz4FrameHipCompressor(int device)
    {
        HipNative.verifyEnabled();
        this.context = new HipNative.Context(device);
    }

    public static boolean isEnabled()
    {
        return HipNative.isEnabled();
    }

    @Override
    public int maxCompressedLength(int uncompressedSize)
    {
        return HipNative.lz4FrameMaxCompressedLength(uncompressedSize);
    }

    @Override
    public int compress(byte[] input, int inputOffset, int inputLength, byte[] output, int outputOffset, int maxOutputLength)
    {
        verifyRange(input, inputOffset, inputLength);
        verifyRange(output, outputOffset, maxOutputLength);
        MemorySegment inputSegment = MemorySegment.ofArray(input).asSlice(inputOffset, inputLength);
        MemorySegment outputSegment = MemorySegment.ofArray(output).asSlice(outputOffset, maxOutputLength);
        return context.singleBlock(HipNative.OP_LZ4FRAME_COMPRESS, inputSegment, inputLength, outputSegment, maxOutputLength);
    }

    @Override
    public int compress(MemorySegment input, MemorySegment output)
    {
        return context.singleBlock(HipNative.OP_LZ4FRAME_COMPRESS, input, toIntExact(input.byteSize()), output, toIntExact(output.byteSize()));
    }

    private static void verifyRange(byte[] data, int offset, int length)
    {
        requireNonNull(data, "data is null");
        if (offset < 0 || length < 0 || offset + length > data.length) {
            throw new IllegalArgumentException(format("Invalid offset or length (%s, %s) in array of length %s", offset, length, data.length));
        }
    }
}
