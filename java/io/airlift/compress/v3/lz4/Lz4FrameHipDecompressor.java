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

import io.airlift.compress.v3.Decompressor;
import io.airlift.compress.v3.MalformedInputException;
import io.airlift.compress.v3.hip.HipNative;

import java.lang.foreign.MemorySegment;

import static java.lang.Math.toIntExact;
import static java.lang.String.format;
import static java.util.Objects.requireNonNull;

/**
 * LZ4 frame decompressor running on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}.
 * Drop-in for {@code Lz4FrameJavaDecompressor}: concatenated and skippable frames, header / block / content checksums;
 * the kernel runs the loop of {@code Lz4FrameCompression.decompress} with its checks in its order, so corrupt input raises
 * the same {@link MalformedInputException} (reason and offset), including those of the block decoder underneath.
 * Binding: {@code achip_lz4frame_decompress} (include/aircompressor_hip.h).
 * <p>
 * Not thread-safe (owns one HIP stream).  For throughput use {@link io.airlift.compress.v3.hip.HipBatchCodec}.
 */
public final class Lz4FrameHipDecompressor
        implements Lz4FrameDecompressor
{
    private final HipNative.Context context;

    public Lz4FrameHipDecompressor()
    {
        this(0);
    }

    public Lz4FrameHipDecompressor(int device)
    {
        HipNative.verifyEnabled();
        this.context = new HipNative.Context(device);
    }

    public static boolean isEnabled()
    {
        return HipNative.isEnabled();
    }

    @Override
    public int decompress(byte[] input, int inputOffset, int inputLength, byte[] output, int outputOffset, int maxOutputLength)
            throws MalformedInputException
    {
        verifyRange(input, inputOffset, inputLength);
        verifyRange(output, outputOffset, maxOutputLength);
        MemorySegment inputSegment = MemorySegment.ofArray(input).asSlice(inputOffset, inputLength);
        MemorySegment outputSegment = MemorySegment.ofArray(output).asSlice(outputOffset, maxOutputLength);
        return context.singleBlock(HipNative.OP_LZ4FRAME_DECOMPRESS, inputSegment, inputLength, outputSegment, maxOutputLength);
    }

    @Override
    public int decompress(MemorySegment input, MemorySegment output)
            throws MalformedInputException
    {
        return context.singleBlock(HipNative.OP_LZ4FRAME_DECOMPRESS, input, toIntExact(input.byteSize()), output, toIntExact(output.byteSize()));
    }

    private static void verifyRange(byte[] data, int offset, int length)
    {
        requireNonNull(data, "data is null");
        if (offset < 0 || length < 0 || offset + length > data.length) {
            throw new IllegalArgumentException(format("Invalid offset or length (%s, %s) in array of length %s", offset, length, data.length));
        }
    }
}
