"""Python mirror of the reference's block codec interfaces for the HIP backend.

    Compressor   -- M/Compressor.java:18-36
    Decompressor -- M/Decompressor.java:18-31

`Lz4HipCompressor` etc. are what `java/io/airlift/compress/v3/{lz4,snappy,zstd}/*Hip*.java`
are in Java: same method names (snake_case), same argument meaning, same exceptions.  The
byte[] overloads take (buffer, offset, length, ...) exactly like the Java ones; the
MemorySegment overloads take two buffer-protocol objects (`memoryview` = MemorySegment).
"""
import ctypes

import numpy as np

from . import native
from .errors import IllegalArgumentException
from .native import HipNative


def _verify_range(data, offset, length):
    # M/lz4/Lz4JavaCompressor.java:78-84
    if data is None:
        raise TypeError("data is null")
    n = len(data)
    if offset < 0 or length < 0 or offset + length > n:
        raise IllegalArgumentException("Invalid offset or length (%s, %s) in array of length %s" % (offset, length, n))


def _ro_view(buf):
    a = np.frombuffer(buf, dtype=np.uint8)
    return a


def _rw_view(buf):
    mv = memoryview(buf)
    if mv.readonly:
        raise IllegalArgumentException("MemorySegment is read-only")  # M/lz4/UnsafeUtil.java:53-55
    return np.frombuffer(mv, dtype=np.uint8)


class Compressor:
    def max_compressed_length(self, uncompressed_size):
        raise NotImplementedError

    def compress(self, input, input_offset, input_length, output, output_offset, max_output_length):
        raise NotImplementedError

    def compress_segment(self, input, output):
        raise NotImplementedError

    def get_retained_size_in_bytes(self, input_length):
        return 0


class Decompressor:
    def decompress(self, input, input_offset, input_length, output, output_offset, max_output_length):
        raise NotImplementedError

    def decompress_segment(self, input, output):
        raise NotImplementedError


class _HipCodecBase:
    _codec = None

    def __init__(self, device=0, native_ctx=None):
        self._native = native_ctx if native_ctx is not None else HipNative(device)
        self._lib = self._native.lib

    @staticmethod
    def is_enabled():
        return HipNative.is_enabled()

    def _call(self, fn_name, src, dst):
        fn = getattr(self._lib, fn_name)
        eo = ctypes.c_int64(0)
        sp = src.ctypes.data if src.size else None
        dp = dst.ctypes.data if dst.size else None
        r = fn(self._native.ctx, sp, dp, int(src.size), int(dst.size), ctypes.byref(eo))
        return r, eo.value


class _HipCompressor(_HipCodecBase, Compressor):
    def max_compressed_length(self, uncompressed_size):
        return getattr(self._lib, "achip_%s_max_compressed_length" % self._codec)(uncompressed_size)

    def compress(self, input, input_offset, input_length, output, output_offset, max_output_length):
        _verify_range(input, input_offset, input_length)
        _verify_range(output, output_offset, max_output_length)
        src = _ro_view(input)[input_offset:input_offset + input_length]
        dst = _rw_view(output)[output_offset:output_offset + max_output_length]
        return self._compress(src, dst)

    def compress_segment(self, input, output):
        return self._compress(_ro_view(input), _rw_view(output))

    def _compress(self, src, dst):
        r, eo = self._call("achip_%s_compress" % self._codec, src, dst)
        if r < 0:
            native.raise_for_status(r, eo)
        return r


class _HipDecompressor(_HipCodecBase, Decompressor):
    def decompress(self, input, input_offset, input_length, output, output_offset, max_output_length):
        _verify_range(input, input_offset, input_length)
        _verify_range(output, output_offset, max_output_length)
        src = _ro_view(input)[input_offset:input_offset + input_length]
        dst = _rw_view(output)[output_offset:output_offset + max_output_length]
        return self._decompress(src, dst)

    def decompress_segment(self, input, output):
        return self._decompress(_ro_view(input), _rw_view(output))

    def _decompress(self, src, dst):
        r, eo = self._call("achip_%s_decompress" % self._codec, src, dst)
        if r < 0:
            native.raise_for_status(r, eo)
        return r


class Lz4HipCompressor(_HipCompressor):
    """Drop-in for Lz4JavaCompressor (M/lz4/Lz4JavaCompressor.java:29-85)."""
    _codec = "lz4"

    def get_retained_size_in_bytes(self, input_length):
        # Lz4RawCompressor.computeTableSize  M/lz4/Lz4RawCompressor.java:304-311
        target = 0 if input_length <= 1 else (1 << ((input_length - 1).bit_length() - 1)) << 1
        return max(16, min(4096, target))


class Lz4HipDecompressor(_HipDecompressor):
    """Drop-in for Lz4JavaDecompressor (M/lz4/Lz4JavaDecompressor.java:28-78)."""
    _codec = "lz4"

    def _decompress(self, src, dst):
        r, eo = self._call("achip_lz4_decompress", src, dst)
        if r < 0:
            if native.status_detail(r) == native.DETAIL_LZ4_EMPTY_OUTPUT:
                return -1  # the Java method returns -1 here (M/lz4/Lz4RawDecompressor.java:52-57)
            native.raise_for_status(r, eo)
        return r


class Lz4FrameHipCompressor(_HipCompressor):
    """Drop-in for Lz4FrameJavaCompressor (M/lz4/Lz4FrameJavaCompressor.java:26-44 -> Lz4FrameCompression.compress): one LZ4
    frame of independent 4 MiB blocks, each through the HIP block encoder, stored uncompressed when that is not smaller."""
    _codec = "lz4frame"

    def max_compressed_length(self, uncompressed_size):
        r = self._lib.achip_lz4frame_max_compressed_length(uncompressed_size)
        if r < 0:
            raise IllegalArgumentException("uncompressedSize is negative: %d" % uncompressed_size if uncompressed_size < 0 else
                                           "Maximum compressed length exceeds Integer.MAX_VALUE for uncompressedSize: %d" % uncompressed_size)
        return r


class Lz4FrameHipDecompressor(_HipDecompressor):
    """Drop-in for Lz4FrameJavaDecompressor (M/lz4/Lz4FrameJavaDecompressor.java:26-44 -> Lz4FrameCompression.decompress):
    concatenated and skippable frames, header / block / content checksums, the reference's exception for every malformed input."""
    _codec = "lz4frame"


class SnappyFramedHipCompressor(_HipCompressor):
    """One-shot form of SnappyFramedOutputStream (M/snappy/SnappyFramedOutputStream.java:73-96, 113-145, 200-255): what
    `new SnappyFramedOutputStream(c, out); write(data); close()` leaves in `out` -- the stream header, then per 64 KiB block a
    chunk with the masked CRC-32C of its plaintext, Snappy-compressed by the HIP block encoder or stored raw when it does not
    reach 0.85.  max_compressed_length is the capacity this form asks for (the stream class has none)."""
    _codec = "snappyframed"

    def max_compressed_length(self, uncompressed_size):
        r = self._lib.achip_snappyframed_max_compressed_length(uncompressed_size)
        if r < 0:
            raise IllegalArgumentException("uncompressedSize is negative: %d" % uncompressed_size if uncompressed_size < 0 else
                                           "Maximum compressed length exceeds Integer.MAX_VALUE for uncompressedSize: %d" % uncompressed_size)
        return r


class SnappyFramedHipDecompressor(_HipDecompressor):
    """One-shot form of SnappyFramedInputStream read to the end of the stream (M/snappy/SnappyFramedInputStream.java:52-73,
    135-305; checksums verified): compressed, raw, skippable and stream-identifier chunks, the reference's error for every
    malformed stream (its IOException / EOFException texts are the ACHIP_D_SNF_* messages)."""
    _codec = "snappyframed"


class _HadoopHipCompressor(_HipCompressor):
    _codec_id = 0

    def __init__(self, device=0, native_ctx=None, buffer_size=262144):
        super().__init__(device, native_ctx)
        self.buffer_size = buffer_size

    def max_compressed_length(self, uncompressed_size):
        r = self._lib.achip_hadoop_max_compressed_length(self._codec_id, uncompressed_size, self.buffer_size)
        if r < 0:
            raise IllegalArgumentException("uncompressedSize is negative: %d" % uncompressed_size if uncompressed_size < 0 else
                                           "Maximum compressed length exceeds Integer.MAX_VALUE for uncompressedSize: %d" % uncompressed_size)
        return r

    def _compress(self, src, dst):
        self._native.set_option("hadoop.buffer_size", self.buffer_size)
        return super()._compress(src, dst)


class _HadoopHipDecompressor(_HipDecompressor):
    def __init__(self, device=0, native_ctx=None, buffer_size=262144):
        super().__init__(device, native_ctx)
        self.buffer_size = buffer_size

    def _decompress(self, src, dst):
        self._native.set_option("hadoop.buffer_size", self.buffer_size)
        return super()._decompress(src, dst)


class Lz4HadoopHipCompressor(_HadoopHipCompressor):
    """One-shot form of Lz4HadoopOutputStream (M/lz4/Lz4HadoopOutputStream.java:60-118; the stream behind
    org.apache.hadoop.io.compress.Lz4Codec, M/lz4/Lz4HadoopStreams.java:52-66): what `createOutputStream(out); write(data); close()`
    leaves in `out` (T/HadoopCodecCompressor.java:57-72) -- per chunk of buffer_size - max(buffer_size / 100, 10) bytes a big-endian
    plaintext length, a big-endian compressed length and the LZ4 block of the HIP encoder."""
    _codec = "lz4hadoop"
    _codec_id = 0


class Lz4HadoopHipDecompressor(_HadoopHipDecompressor):
    """One-shot form of Lz4HadoopInputStream read to the end (M/lz4/Lz4HadoopInputStream.java:47-156, driven as
    T/HadoopCodecDecompressor.java:40-60 does): blocks of several chunks, empty blocks, the stream's own bufferSize + 8 byte buffer
    when the destination has less room than a block declares; its IOException / EOFException texts are the ACHIP_D_HDP_* messages."""
    _codec = "lz4hadoop"


class SnappyHadoopHipCompressor(_HadoopHipCompressor):
    """One-shot form of SnappyHadoopOutputStream (M/snappy/SnappyHadoopOutputStream.java:60-131): chunks of
    buffer_size - (buffer_size / 6 + 32) plaintext bytes."""
    _codec = "snappyhadoop"
    _codec_id = 1


class SnappyHadoopHipDecompressor(_HadoopHipDecompressor):
    """One-shot form of SnappyHadoopInputStream read to the end (M/snappy/SnappyHadoopInputStream.java:44-170)."""
    _codec = "snappyhadoop"


class SnappyHipCompressor(_HipCompressor):
    """Drop-in for SnappyJavaCompressor (M/snappy/SnappyJavaCompressor.java:26-91)."""
    _codec = "snappy"


class SnappyHipDecompressor(_HipDecompressor):
    """Drop-in for SnappyJavaDecompressor (M/snappy/SnappyJavaDecompressor.java:28-89)."""
    _codec = "snappy"

    def get_uncompressed_length(self, compressed, compressed_offset):
        src = _ro_view(compressed)[compressed_offset:]
        eo = ctypes.c_int64(0)
        r = self._lib.achip_snappy_uncompressed_length(src.ctypes.data if src.size else None, int(src.size), ctypes.byref(eo))
        if r < 0:
            native.raise_for_status(int(r), eo.value)
        return int(r)


class ZstdHipCompressor(_HipCompressor):
    """Drop-in for ZstdJavaCompressor (always level 3; M/zstd/ZstdJavaCompressor.java:28-90)."""
    _codec = "zstd"


class ZstdHipOutputStream:
    """Drop-in for ZstdOutputStream (M/zstd/ZstdOutputStream.java:30-221) over a binary sink, INCREMENTAL like the Java stream: write() hands
    the bytes to the library's stream state (achip_zstdstream_compress_begin / _feed / _finish), which keeps the Java stream's 4 MiB buffer on the
    device, flushes whole blocks to the sink whenever that buffer is full (the window slides) and writes the rest and the checksum at close().
    The bytes are the Java stream's whatever the sizes of the write() calls; memory per open stream is a constant (~12 MB of device memory)."""

    def __init__(self, sink, device=0, native_ctx=None):
        self._sink = sink
        self._native = native_ctx if native_ctx is not None else native.HipNative(device)
        self._lib = self._native.lib
        self._state = self._lib.achip_zstdstream_compress_begin(self._native.ctx)
        if not self._state:
            raise native.HipUnavailableError("achip_zstdstream_compress_begin failed: %s" % self._lib.achip_last_error().decode())
        self._out = np.zeros(1 << 20, dtype=np.uint8)
        self._closed = False

    def write(self, buffer, offset=0, length=None):
        if self._closed:
            raise IOError("Stream is closed")  # :72-74
        view = _ro_view(buffer)
        length = view.size - offset if length is None else length
        _verify_range(buffer, offset, length)
        data = np.ascontiguousarray(view[offset:offset + length])
        consumed, produced = ctypes.c_int64(0), ctypes.c_int64(0)
        at = 0
        while True:
            r = self._lib.achip_zstdstream_compress_feed(self._native.ctx, self._state, data[at:].ctypes.data if at < length else None, int(length - at), self._out.ctypes.data,
                                                         int(self._out.size), ctypes.byref(consumed), ctypes.byref(produced))
            if r < 0:
                native.raise_for_status(int(r))
            if produced.value:
                self._sink.write(self._out[:produced.value].tobytes())
            at += consumed.value
            if at >= length and produced.value < self._out.size:
                return

    def close(self):
        if self._closed:
            return
        # (ZstdOutputStream.close :193-205 sets `closed` only behind writeChunk(true); a failing encode leaves the stream open -- and the sink is
        # closed either way, as its try / finally does)
        try:
            produced = ctypes.c_int64(0)
            while True:
                r = self._lib.achip_zstdstream_compress_finish(self._native.ctx, self._state, self._out.ctypes.data, int(self._out.size), ctypes.byref(produced))
                if r < 0:
                    native.raise_for_status(int(r))
                if produced.value:
                    self._sink.write(self._out[:produced.value].tobytes())
                if r == 1:
                    break
            self._closed = True
            self._lib.achip_zstdstream_compress_end(self._native.ctx, self._state)
            self._state = None
        finally:
            if hasattr(self._sink, "close"):
                self._sink.close()

    def __del__(self):
        try:
            if getattr(self, "_state", None):
                self._lib.achip_zstdstream_compress_end(self._native.ctx, self._state)
                self._state = None
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _ZstdStreamEncoder(_HipCompressor):
    _codec = "zstdstream"


class ZstdHipInputStream:
    """Drop-in for ZstdInputStream (M/zstd/ZstdInputStream.java:28-151 over ZstdIncrementalFrameDecompressor.java:44-386) over a binary source,
    INCREMENTAL like the Java stream: input is read from the source a megabyte at a time and handed to the library's stream state
    (achip_zstdstream_decompress_begin / _feed / _end), which decodes a frame in steps of whole blocks with tables, repeat offsets, window and
    running checksum carried on the device -- ~45 MB of host + device memory per open stream at the Java writer's 8 MiB window, whatever the
    stream's length (a frame of gigabytes, many frames back to back).  A damaged stream delivers every byte in front of the damaged block and
    fails at the read that reaches it, where ZstdInputStream.read throws; the end of the source anywhere but between frames is "Not enough
    input bytes" (ZstdInputStream.java:79-85).  `max_decoded_bytes` (None = no limit, the default) makes the stream fail with an IOError once
    it has delivered more than that -- a guard for callers who collect the whole plaintext (a few KB of RLE block headers decode to
    gigabytes); the stream itself no longer needs it."""

    DEFAULT_MAX_DECODED_BYTES = None
    READ_SIZE = 1 << 20

    def __init__(self, source, device=0, native_ctx=None, max_decoded_bytes=DEFAULT_MAX_DECODED_BYTES):
        if max_decoded_bytes is not None and max_decoded_bytes < 0:
            raise ValueError("max_decoded_bytes must be >= 0 or None")
        self._source = source
        self._max_decoded = max_decoded_bytes
        self._native = native_ctx if native_ctx is not None else native.HipNative(device)
        self._lib = self._native.lib
        self._state = self._lib.achip_zstdstream_decompress_begin(self._native.ctx)
        if not self._state:
            raise native.HipUnavailableError("achip_zstdstream_decompress_begin failed: %s" % self._lib.achip_last_error().decode())
        self._input = np.zeros(0, dtype=np.uint8)  # read from the source, not yet taken by the library
        self._source_ended = False
        self._delivered = 0
        self._seen_input = False
        self._closed = False

    def _more_input(self):
        if self._source_ended:
            return False
        data = self._source.read(self.READ_SIZE)
        if not data:
            self._source_ended = True
            return False
        self._seen_input = True
        self._input = np.frombuffer(bytes(data), dtype=np.uint8)
        return True

    def read_into(self, output_buffer, output_offset, output_length):
        """ZstdInputStream.read(byte[], int, int) :63-105: the number of bytes delivered (the buffer is filled unless the stream ends), -1 at the
        end of the stream"""
        if self._closed:
            raise IOError("Stream is closed")  # :66-68
        _verify_range(output_buffer, output_offset, output_length)
        if output_length == 0:
            return 0
        out = _rw_view(output_buffer)[output_offset:output_offset + output_length]
        used = 0
        consumed, produced, err = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        while used < output_length:
            r = self._lib.achip_zstdstream_decompress_feed(self._native.ctx, self._state, self._input.ctypes.data if self._input.size else None, int(self._input.size),
                                                           out[used:].ctypes.data, int(output_length - used), ctypes.byref(consumed), ctypes.byref(produced), ctypes.byref(err))
            if r < 0:
                if used > 0:
                    break  # (what was decoded goes out; the next read fails)
                native.raise_for_status(int(r), err.value)
            self._input = self._input[consumed.value:]
            used += produced.value
            if produced.value == 0 and self._input.size == 0 and not self._more_input():
                # the source has ended: between frames that is the end of the stream, anywhere else the stream is cut short
                if self._lib.achip_zstdstream_decompress_at_stopping_point(self._state) and self._seen_input:
                    break
                if used > 0:
                    break
                raise IOError("Not enough input bytes")
        self._delivered += used
        if self._max_decoded is not None and self._delivered > self._max_decoded:
            raise IOError("Decoded size %d exceeds max_decoded_bytes %d" % (self._delivered, int(self._max_decoded)))
        return used if used > 0 else -1

    def read(self, n=-1):
        """io-style: up to n bytes (all that is left for n < 0), b"" at the end"""
        if self._closed:
            raise IOError("Stream is closed")
        pieces = []
        want = None if n is None or n < 0 else int(n)
        buf = bytearray(1 << 20 if want is None else max(1, min(want, 1 << 20)))
        while want is None or want > 0:
            k = self.read_into(buf, 0, len(buf) if want is None else min(len(buf), want))
            if k < 0:
                break
            pieces.append(bytes(buf[:k]))
            if want is not None:
                want -= k
        return b"".join(pieces)

    def close(self):
        if not self._closed:
            self._closed = True
            if self._state:
                self._lib.achip_zstdstream_decompress_end(self._native.ctx, self._state)
                self._state = None
            if hasattr(self._source, "close"):
                self._source.close()

    def __del__(self):
        try:
            if getattr(self, "_state", None):
                self._lib.achip_zstdstream_decompress_end(self._native.ctx, self._state)
                self._state = None
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ZstdHipDecompressor(_HipDecompressor):
    """Drop-in for ZstdJavaDecompressor (M/zstd/ZstdJavaDecompressor.java:29-90)."""
    _codec = "zstd"

    def get_decompressed_size(self, input, offset, length):
        src = _ro_view(input)[offset:offset + length]
        eo = ctypes.c_int64(0)
        r = self._lib.achip_zstd_decompressed_size(src.ctypes.data if src.size else None, int(src.size), ctypes.byref(eo))
        if r < -1:
            native.raise_for_status(int(r), eo.value)
        return int(r)
