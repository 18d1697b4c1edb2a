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
import java.io.InputStream;
import java.lang.foreign.MemorySegment;

import static java.util.Objects.checkFromIndexSize;
import static java.util.Objects.requireNonNull;

/**
 * {@code ZstdInputStream} with the decoder on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}, INCREMENTAL like the Java
 * stream ({@code ZstdInputStream.java:63-105} over {@code ZstdIncrementalFrameDecompressor.java:44-72,216-234}): input is read from the
 * underlying stream a megabyte at a time and handed to the library's stream state ({@code achip_zstdstream_decompress_begin / _feed / _end}),
 * which decodes a frame in steps of whole blocks with tables, repeat offsets, window and running checksum carried on the device -- about
 * 45 MB of host and device memory per open stream at {@code ZstdOutputStream}'s window, whatever the stream's length (one frame of
 * gigabytes, many frames back to back).  A damaged stream delivers every byte in front of the damaged block and fails at the read that
 * reaches it; the end of the input anywhere but between frames is "Not enough input bytes".
 * To read many streams at once use {@link io.airlift.compress.v3.hip.HipBatchCodec} with {@link HipNative#OP_ZSTD_DECOMPRESS}: one item per
 * stream, capacities from {@link HipNative#zstdDecompressBound}.
 */
public final class ZstdHipInputStream
        extends InputStream
{
    private static final int READ_SIZE = 1 << 20;

    private final InputStream inputStream;
    private final HipNative.Context context;
    private final HipNative.Context.ZstdDecodeStream decoder;
    private final byte[] input = new byte[READ_SIZE];
    private int inputOffset;
    private int inputLimit;
    private boolean inputEnded;
    private boolean sawInput;
    private byte[] singleByte;
    private boolean closed;

    public ZstdHipInputStream(InputStream inputStream)
    {
        this(inputStream, 0);
    }

    public ZstdHipInputStream(InputStream inputStream, int device)
    {
        this.inputStream = requireNonNull(inputStream, "inputStream is null");
        HipNative.verifyEnabled();
        this.context = new HipNative.Context(device);
        HipNative.Context.ZstdDecodeStream opened;
        try {
            opened = context.openZstdDecodeStream();
        }
        catch (RuntimeException | Error e) {
            context.close();  // (the native context must not outlive a constructor that failed)
            throw e;
        }
        this.decoder = opened;
    }

    @Override
    public int read()
            throws IOException
    {
        if (singleByte == null) {
            singleByte = new byte[1];
        }
        return read(singleByte, 0, 1) == 1 ? singleByte[0] & 0xFF : -1;
    }

    @Override
    public int read(byte[] outputBuffer, int outputOffset, int outputLength)
            throws IOException
    {
        if (closed) {
            throw new IOException("Stream is closed");
        }
        checkFromIndexSize(outputOffset, outputLength, outputBuffer.length);
        if (outputLength == 0) {
            return 0;
        }
        // (As ZstdInputStream.read :79-103 this fills the caller's buffer unless the stream ends -- the reference loops `while (outputUsed < outputLength)`
        // and asks its source for more in between; a reader that returned at the first output would deliver the same bytes in other portions.)
        int used = 0;
        while (used < outputLength) {
            try {
                decoder.feed(
                        MemorySegment.ofArray(input).asSlice(inputOffset), inputLimit - inputOffset,
                        MemorySegment.ofArray(outputBuffer).asSlice(outputOffset + used), outputLength - used);
            }
            catch (RuntimeException e) {
                if (used > 0) {
                    break;  // what was decoded goes out; the next read fails
                }
                throw e;
            }
            inputOffset += (int) decoder.consumed();
            int produced = (int) decoder.produced();
            used += produced;
            if (produced == 0 && inputOffset == inputLimit && !fill()) {
                // the input has ended: between frames that is the end of the stream, anywhere else the stream is cut short (ZstdInputStream.java:79-85)
                if ((decoder.atStoppingPoint() && sawInput) || used > 0) {
                    break;
                }
                throw new IOException("Not enough input bytes");
            }
        }
        return used > 0 ? used : -1;
    }

    private boolean fill()
            throws IOException
    {
        if (inputEnded) {
            return false;
        }
        int size = inputStream.read(input, 0, input.length);
        if (size <= 0) {
            inputEnded = size < 0;
            inputOffset = 0;
            inputLimit = 0;
            return false;
        }
        sawInput = true;
        inputOffset = 0;
        inputLimit = size;
        return true;
    }

    @Override
    public void close()
            throws IOException
    {
        if (!closed) {
            closed = true;
            try {
                decoder.close();
                context.close();
            }
            finally {
                inputStream.close();
            }
        }
    }
}
