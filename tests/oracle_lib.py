"""ctypes view of oracle/liboracle.so -- the CPU restatement of the reference's Java codecs.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by aircompressor_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

OP_LZ4_DECOMPRESS, OP_LZ4_COMPRESS, OP_SNAPPY_DECOMPRESS, OP_SNAPPY_COMPRESS, OP_ZSTD_DECOMPRESS, OP_ZSTD_COMPRESS = range(6)


def status_class(status):
    return (-status) & 15 if status < 0 else 0


def status_detail(status):
    return (-status) >> 4 if status < 0 else 0


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        u8p = ctypes.c_void_p
        i64 = ctypes.c_int64
        for name in ("orc_lz4_max_compressed_length", "orc_snappy_max_compressed_length", "orc_zstd_max_compressed_length", "orc_lz4frame_max_compressed_length", "orc_snappyframed_max_compressed_length"):
            getattr(lib, name).restype = i64
            getattr(lib, name).argtypes = [i64]
        lib.orc_zstd_stream_max_compressed_length.restype = i64
        lib.orc_zstd_stream_max_compressed_length.argtypes = [i64]
        for name in ("orc_lz4_compress", "orc_snappy_compress", "orc_zstd_compress", "orc_zstd_stream_compress", "orc_lz4frame_compress", "orc_snappyframed_compress"):
            getattr(lib, name).restype = i64
            getattr(lib, name).argtypes = [u8p, i64, u8p, i64]
        for name in ("orc_lz4_decompress", "orc_snappy_decompress", "orc_zstd_decompress", "orc_lz4frame_decompress", "orc_snappyframed_decompress"):
            getattr(lib, name).restype = i64
            getattr(lib, name).argtypes = [u8p, i64, u8p, i64, ctypes.POINTER(i64)]
        for name in ("orc_snappy_uncompressed_length", "orc_zstd_decompressed_size"):
            getattr(lib, name).restype = i64
            getattr(lib, name).argtypes = [u8p, i64, ctypes.POINTER(i64)]
        lib.orc_xxh64.restype = ctypes.c_uint64
        lib.orc_xxh64.argtypes = [u8p, i64, ctypes.c_uint64]
        lib.orc_xxh32.restype = ctypes.c_uint32
        lib.orc_xxh32.argtypes = [u8p, i64, ctypes.c_uint32]
        for name in ("orc_crc32c", "orc_masked_crc32c"):
            getattr(lib, name).restype = ctypes.c_uint32
            getattr(lib, name).argtypes = [u8p, i64]
        lib.orc_hadoop_max_compressed_length.restype = i64
        lib.orc_hadoop_max_compressed_length.argtypes = [ctypes.c_int32, i64, ctypes.c_int32]
        lib.orc_hadoop_compress.restype = i64
        lib.orc_hadoop_compress.argtypes = [ctypes.c_int32, u8p, i64, u8p, i64, ctypes.c_int32]
        lib.orc_hadoop_decompress.restype = i64
        lib.orc_hadoop_decompress.argtypes = [ctypes.c_int32, u8p, i64, u8p, i64, ctypes.c_int32, ctypes.POINTER(i64)]
        lib.orc_random_generator.restype = None
        lib.orc_random_generator.argtypes = [ctypes.c_double, u8p, i64]
        lib.orc_batch.restype = i64
        lib.orc_batch.argtypes = [ctypes.c_int32] + [u8p] * 9 + [ctypes.c_int32]

    @staticmethod
    def _buf(data):
        arr = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        if arr.size == 0:
            arr = np.zeros(1, dtype=np.uint8)[:0]
        return arr

    def max_compressed_length(self, codec, n):
        return getattr(self.lib, "orc_%s_max_compressed_length" % codec)(n)

    def compress(self, codec, data, cap=None):
        """returns bytes, or raises OracleError"""
        src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        if cap is None:
            cap = self.max_compressed_length(codec, len(src))
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        r = getattr(self.lib, "orc_%s_compress" % codec)(src.ctypes.data if len(src) else None, len(src), dst.ctypes.data, cap)
        if r < 0:
            raise OracleError(r, 0)
        return dst[:r].tobytes()

    def zstd_stream_compress(self, data, cap=None):
        """what ZstdOutputStream puts on its sink for write(data) + close()"""
        src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        if cap is None:
            cap = self.lib.orc_zstd_stream_max_compressed_length(len(src))
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        r = self.lib.orc_zstd_stream_compress(src.ctypes.data if len(src) else None, len(src), dst.ctypes.data, cap)
        if r < 0:
            raise OracleError(r, 0)
        return dst[:r].tobytes()

    def decompress(self, codec, data, cap):
        src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        eo = ctypes.c_int64(0)
        r = getattr(self.lib, "orc_%s_decompress" % codec)(src.ctypes.data if len(src) else None, len(src), dst.ctypes.data, cap, ctypes.byref(eo))
        if r < 0:
            raise OracleError(r, eo.value)
        return dst[:r].tobytes()

    # Hadoop LZ4 / Snappy block streams (oracle/hadoop_streams.c); codec "lz4" | "snappy"
    def hadoop_max_compressed_length(self, codec, n, buffer_size=262144):
        return self.lib.orc_hadoop_max_compressed_length(0 if codec == "lz4" else 1, n, buffer_size)

    def hadoop_compress(self, codec, data, buffer_size=262144, cap=None):
        src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        if cap is None:
            cap = self.hadoop_max_compressed_length(codec, len(src), buffer_size)
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        r = self.lib.orc_hadoop_compress(0 if codec == "lz4" else 1, src.ctypes.data if len(src) else None, len(src), dst.ctypes.data, cap, buffer_size)
        if r < 0:
            raise OracleError(r, 0)
        return dst[:r].tobytes()

    def hadoop_decompress(self, codec, data, cap, buffer_size=262144):
        src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        eo = ctypes.c_int64(0)
        r = self.lib.orc_hadoop_decompress(0 if codec == "lz4" else 1, src.ctypes.data if len(src) else None, len(src), dst.ctypes.data, cap, buffer_size, ctypes.byref(eo))
        if r < 0:
            raise OracleError(r, eo.value)
        return dst[:r].tobytes()

    def xxh32(self, data, seed=0):
        src = np.frombuffer(bytes(data), dtype=np.uint8)
        return self.lib.orc_xxh32(src.ctypes.data if len(src) else None, len(src), seed & 0xFFFFFFFF)

    def crc32c(self, data, masked=False):
        src = np.frombuffer(bytes(data), dtype=np.uint8)
        return getattr(self.lib, "orc_masked_crc32c" if masked else "orc_crc32c")(src.ctypes.data if len(src) else None, len(src))

    def xxh64(self, data, seed=0):
        src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        return self.lib.orc_xxh64(src.ctypes.data if len(src) else None, len(src), seed)

    def random_generator(self, ratio, n=1048576 + 100):
        out = np.zeros(n, dtype=np.uint8)
        self.lib.orc_random_generator(ratio, out.ctypes.data, n)
        return out

    def batch(self, op, src, src_off, src_len, dst, dst_off, dst_cap):
        """numpy arrays in, (out_len, status, err_off, total) out"""
        n = len(src_off)
        out_len = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        err_off = np.zeros(n, dtype=np.int64)
        total = self.lib.orc_batch(op, src.ctypes.data, src_off.ctypes.data, src_len.ctypes.data, dst.ctypes.data, dst_off.ctypes.data,
                                   dst_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data, err_off.ctypes.data, n)
        return out_len, status, err_off, total


class OracleError(Exception):
    def __init__(self, status, offset):
        super().__init__("oracle status=%d class=%d detail=%d offset=%d" % (status, status_class(status), status_detail(status), offset))
        self.status = status
        self.cls = status_class(status)
        self.detail = status_detail(status)
        self.offset = offset


_cached = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load():
    global _cached
    if _cached is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build()
        if os.environ.get("ORACLE_LIB"):  # (e.g. liboracle_asan.so -- `make -C oracle asan`, run with the sanitizer runtime preloaded)
            path = os.path.join(ORACLE_DIR, os.environ["ORACLE_LIB"])
        _cached = Oracle(ctypes.CDLL(path))
    return _cached
