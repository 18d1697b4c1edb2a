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

import java.io.ByteArrayOutputStream;
import java.io.IOException;
import java.io.OutputStream;
import java.lang.foreign.MemorySegment;

import static java.util.Objects.requireNonNull;

/**
 * {@code ZstdOutputStream} with the encoder on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}: what is written is
 * collected and, at {@code close()}, handed to {@code achip_zstdstream_compress}, which produces what {@code ZstdOutputStream} puts on its
 * sink for the same bytes -- the stream's parameters (those for an unknown input size: window 2^20 whatever the size), not
 * {@code ZstdHipCompressor}'s -- and that frame goes to the sink.
 * <p>
 * {@code ZstdOutputStream} starts flushing chunks once 4 MiB have been written (a frame header without the content size, the window
 * slid between chunks); the library writes those bytes too -- they only reach the sink in one piece at {@code close()} -- up to
 * 2^30 bytes per stream.  To write many streams at once use {@link io.airlift.compress.v3.hip.HipBatchCodec} with
 * {@link HipNative#OP_ZSTDSTREAM_COMPRESS}: one item per stream.
 * <p>
 * Reading: {@link ZstdHipInputStream}.
 */
public final class ZstdHipOutputStream
        extends OutputStream
{
    private final OutputStream outputStream;
    private final HipNative.Context context;
    private final ByteArrayOutputStream pending = new ByteArrayOutputStream();
    private boolean closed;

    public ZstdHipOutputStream(OutputStream outputStream)
    {
        this(outputStream, 0);
    }

    public ZstdHipOutputStream(OutputStream outputStream, int device)
    {
        this.outputStream = requireNonNull(outputStream, "outputStream is null");
        HipNative.verifyEnabled();
        this.context = new HipNative.Context(device);
    }

    @Override
    public void write(int b)
            throws IOException
    {
        if (closed) {
            throw new IOException("Stream is closed");
        }
        pending.write(b);
    }

    @Override
    public void write(byte[] buffer, int offset, int length)
            throws IOException
    {
        if (closed) {
            throw new IOException("Stream is closed");
        }
        pending.write(buffer, offset, length);
    }

    @Override
    public void close()
            throws IOException
    {
        if (closed) {
            return;
        }
        // (ZstdOutputStream.close sets `closed` only behind writeChunk(true) and closes its sink in a finally block: M/zstd/ZstdOutputStream.java:193-205)
        try {
            byte[] input = pending.toByteArray();
            byte[] output = new byte[HipNative.zstdStreamMaxCompressedLength(input.length)];
            int size = context.singleBlock(HipNative.OP_ZSTDSTREAM_COMPRESS, MemorySegment.ofArray(input), input.length, MemorySegment.ofArray(output), output.length);
            outputStream.write(output, 0, size);
            closed = true;
            context.close();  // (the native context -- a HIP stream and scratch -- goes with the stream, not with the garbage collector)
        }
        finally {
            outputStream.close();
        }
    }
}
