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

import java.io.IOException;
import java.io.OutputStream;
import java.lang.foreign.MemorySegment;

import static java.util.Objects.requireNonNull;

/**
 * {@code ZstdOutputStream} with the encoder on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}, INCREMENTAL like the Java
 * stream ({@code ZstdOutputStream.java:93-221}): {@code write} hands the bytes to the library's stream state
 * ({@code achip_zstdstream_compress_begin / _feed / _finish}), which keeps the Java stream's 4 MiB buffer on the device, flushes whole blocks to
 * the sink whenever that buffer is full (the window slides) and writes the rest and the checksum at {@code close()}.  The bytes are
 * {@code ZstdOutputStream}'s whatever the sizes of the writes -- the stream's parameters (those for an unknown input size: window 2^20), not
 * {@code ZstdHipCompressor}'s; memory per open stream is a constant.  To write many streams at once use
 * {@link io.airlift.compress.v3.hip.HipBatchCodec} with {@link HipNative#OP_ZSTDSTREAM_COMPRESS}: one item per stream.
 * <p>
 * Reading: {@link ZstdHipInputStream}.
 */
public final class ZstdHipOutputStream
        extends OutputStream
{
    private final OutputStream outputStream;
    private final HipNative.Context context;
    private final HipNative.Context.ZstdEncodeStream encoder;
    private final byte[] flushed = new byte[1 << 20];
    private byte[] singleByte;
    private boolean closed;
    private boolean released;  // the native encoder and context are gone (close() ran, successfully or not)

    public ZstdHipOutputStream(OutputStream outputStream)
    {
        this(outputStream, 0);
    }

    public ZstdHipOutputStream(OutputStream outputStream, int device)
    {
        this.outputStream = requireNonNull(outputStream, "outputStream is null");
        HipNative.verifyEnabled();
        this.context = new HipNative.Context(device);
        this.encoder = context.openZstdEncodeStream();
    }

    @Override
    public void write(int b)
            throws IOException
    {
        if (singleByte == null) {
            singleByte = new byte[1];
        }
        singleByte[0] = (byte) b;
        write(singleByte, 0, 1);
    }

    @Override
    public void write(byte[] buffer, int offset, int length)
            throws IOException
    {
        if (closed || released) {
            throw new IOException("Stream is closed");
        }
        java.util.Objects.checkFromIndexSize(offset, length, buffer.length);
        int at = 0;
        while (true) {
            encoder.feed(MemorySegment.ofArray(buffer).asSlice(offset + at), length - at, MemorySegment.ofArray(flushed), flushed.length);
            int produced = (int) encoder.produced();
            if (produced > 0) {
                outputStream.write(flushed, 0, produced);
            }
            at += (int) encoder.consumed();
            if (at >= length && produced < flushed.length) {
                return;
            }
        }
    }

    @Override
    public void close()
            throws IOException
    {
        if (closed) {
            return;
        }
        // (ZstdOutputStream.close :193-205 sets `closed` only behind writeChunk(true); the sink is closed either way.)  The NATIVE stream state --
        // several MB of device and pinned memory -- and the context are released whatever finish() or the sink's write() do: a close() that threw
        // leaves `closed` false as the reference does, and a second close() finds nothing native left to retry on (ADVICE round 5).
        try {
            if (released) {
                throw new IOException("Stream failed while closing");
            }
            try {
                boolean done;
                do {
                    done = encoder.finish(MemorySegment.ofArray(flushed), flushed.length);
                    int produced = (int) encoder.produced();
                    if (produced > 0) {
                        outputStream.write(flushed, 0, produced);
                    }
                }
                while (!done);
                closed = true;
            }
            finally {
                released = true;
                try {
                    encoder.close();
                }
                finally {
                    context.close();
                }
            }
        }
        finally {
            outputStream.close();
        }
    }
}
