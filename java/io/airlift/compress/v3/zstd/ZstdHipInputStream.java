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
 * {@code ZstdInputStream} with the decoder on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}, in whole-buffer form:
 * the first read takes everything the underlying stream holds, asks {@code achip_zstd_decompress_bound} what the frames can decode to
 * (from their frame and block headers: the frames need NOT carry a content size -- {@code ZstdOutputStream}'s do not from 4 MiB on), decodes
 * all frames in one call and hands the plaintext out as asked.  A damaged stream fails at that first read, not at the read that reaches
 * the damage.  The whole plaintext is held at once, and the bound of a few bytes of input can be huge (every 4-byte RLE block header
 * announces up to 128 KiB): a stream whose bound exceeds {@code maxDecodedBytes} (default {@link #DEFAULT_MAX_DECODED_BYTES}, 1 GiB) is refused
 * with an {@link IOException} before anything is allocated -- {@code ZstdInputStream} decodes the same stream in a window's worth of memory.
 * To read many streams at once use {@link io.airlift.compress.v3.hip.HipBatchCodec} with
 * {@link HipNative#OP_ZSTD_DECOMPRESS}: one item per stream, capacities from {@link HipNative#zstdDecompressBound}.
 */
public final class ZstdHipInputStream
        extends InputStream
{
    public static final long DEFAULT_MAX_DECODED_BYTES = 1L << 30;

    private final InputStream inputStream;
    private final int device;
    private final long maxDecodedBytes;
    private byte[] plain;
    private int position;
    private boolean closed;

    public ZstdHipInputStream(InputStream inputStream)
    {
        this(inputStream, 0);
    }

    public ZstdHipInputStream(InputStream inputStream, int device)
    {
        this(inputStream, device, DEFAULT_MAX_DECODED_BYTES);
    }

    public ZstdHipInputStream(InputStream inputStream, int device, long maxDecodedBytes)
    {
        this.inputStream = requireNonNull(inputStream, "inputStream is null");
        if (maxDecodedBytes < 0) {
            throw new IllegalArgumentException("maxDecodedBytes is negative");
        }
        HipNative.verifyEnabled();
        this.device = device;
        this.maxDecodedBytes = maxDecodedBytes;
    }

    private void fill()
            throws IOException
    {
        if (plain != null) {
            return;
        }
        byte[] input = inputStream.readAllBytes();
        if (input.length == 0) {
            // ZstdInputStream wants a frame magic before it calls the stream ended (ZstdInputStream.java:79-85)
            throw new IOException("Not enough input bytes");
        }
        long bound = HipNative.zstdDecompressBound(MemorySegment.ofArray(input));
        if (bound > maxDecodedBytes) {
            // (memory amplification, not corruption: the frames may well be legal)
            throw new IOException("Decoded size bound " + bound + " exceeds maxDecodedBytes " + maxDecodedBytes);
        }
        if (bound > Integer.MAX_VALUE - 8) {
            throw new IOException("Stream decodes to more than a byte[] holds: " + bound);
        }
        byte[] output = new byte[(int) Math.max(bound, 1)];
        int size = 0;
        if (bound > 0) {
            try (HipNative.Context context = new HipNative.Context(device)) {
                size = context.singleBlock(HipNative.OP_ZSTD_DECOMPRESS, MemorySegment.ofArray(input), input.length, MemorySegment.ofArray(output), (int) bound);
            }
        }
        plain = size == output.length ? output : java.util.Arrays.copyOf(output, size);
    }

    @Override
    public int read()
            throws IOException
    {
        if (closed) {
            throw new IOException("Stream is closed");
        }
        fill();
        return position < plain.length ? plain[position++] & 0xFF : -1;
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
        fill();
        if (position >= plain.length) {
            return -1;
        }
        int size = Math.min(outputLength, plain.length - position);
        System.arraycopy(plain, position, outputBuffer, outputOffset, size);
        position += size;
        return size;
    }

    @Override
    public int available()
    {
        return closed || plain == null ? 0 : plain.length - position;
    }

    @Override
    public void close()
            throws IOException
    {
        if (!closed) {
            closed = true;
            inputStream.close();
        }
    }
}
