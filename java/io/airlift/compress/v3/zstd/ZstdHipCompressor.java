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
package io.airlift.compress.v3.zstd;

import io.airlift.compress.v3.hip.HipNative;

import java.lang.foreign.MemorySegment;

import static java.lang.Math.toIntExact;
import static java.lang.String.format;
import static java.util.Objects.requireNonNull;

/**
 * Zstd block compressor running on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}.
 * Drop-in for {@code ZstdJavaCompressor}: same arguments, same exceptions, and the SAME compressed bytes
 * (the HIP kernel restates the Java encoder's greedy parse step for step).
 * <p>
 * This class is not thread-safe (it owns one HIP stream), like {@code ZstdJavaCompressor}.
 * For throughput use {@link io.airlift.compress.v3.hip.HipBatchCodec}: one call, many blocks, data resident on the device.
 */
public final class ZstdHipCompressor
        implements ZstdCompressor
{
    private final HipNative.Context context;

    public ZstdHipCompressor()
    {
        this(0);
    }

    public ZstdHipCompressor(int device)
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
        return HipNative.zstdMaxCompressedLength(uncompressedSize);
    }

    @Override
    public int compress(byte[] input, int inputOffset, int inputLength, byte[] output, int outputOffset, int maxOutputLength)
    {
        verifyRange(input, inputOffset, inputLength);
        verifyRange(output, outputOffset, maxOutputLength);
        MemorySegment inputSegment = MemorySegment.ofArray(input).asSlice(inputOffset, inputLength);
        MemorySegment outputSegment = MemorySegment.ofArray(output).asSlice(outputOffset, maxOutputLength);
        return context.singleBlock(HipNative.OP_ZSTD_COMPRESS, inputSegment, inputLength, outputSegment, maxOutputLength);
    }

    @Override
    public int compress(MemorySegment input, MemorySegment output)
    {
        return context.singleBlock(HipNative.OP_ZSTD_COMPRESS, input, toIntExact(input.byteSize()), output, toIntExact(output.byteSize()));
    }

    private static void verifyRange(byte[] data, int offset, int length)
    {
        requireNonNull(data, "data is null");
        if (offset < 0 || length < 0 || offset + length > data.length) {
            throw new IllegalArgumentException(format("Invalid offset or length (%s, %s) in array of length %s", offset, length, data.length));
        }
    }
}
