"""ctypes access to the third-party codecs bundled with the reference (build container only).

liblz4 1.10.0 / libsnappy 1.2.1 / libzstd 1.5.6 under
/root/reference/src/main/resources/aircompressor/linux-amd64 (versions: bin/download.sh:57,79,101).
They are an independent *decode* cross-check (a conformant decoder yields identical plaintext) and
a generator of diverse zstd frames; they are NOT a compress oracle.
"""
import ctypes
import os

import numpy as np

LIBDIR = "/root/reference/src/main/resources/aircompressor/linux-amd64"


def available():
    return os.path.isdir(LIBDIR)


_libs = {}


def _load(name, lazy=False):
    if name not in _libs:
        path = os.path.join(LIBDIR, "lib%s.so" % name)
        if lazy:  # liblz4.so has dangling LZ4_XXH32* imports: needs RTLD_LAZY
            h = ctypes._dlopen(path, os.RTLD_LAZY)
            _libs[name] = ctypes.CDLL(path, handle=h)
        else:
            _libs[name] = ctypes.CDLL(path)
    return _libs[name]


def _np(data):
    return np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))


def lz4_compress(data):
    lib = _load("lz4", lazy=True)
    src = _np(data)
    cap = lib.LZ4_compressBound(len(src))
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    lib.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    n = lib.LZ4_compress_default(src.ctypes.data, dst.ctypes.data, len(src), cap)
    assert n > 0 or len(src) == 0
    return dst[:n].tobytes()


def lz4_decompress(data, cap):
    lib = _load("lz4", lazy=True)
    src = _np(data)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    lib.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    n = lib.LZ4_decompress_safe(src.ctypes.data, dst.ctypes.data, len(src), cap)
    if n < 0:
        raise ValueError("LZ4_decompress_safe failed: %d" % n)
    return dst[:n].tobytes()


def snappy_compress(data):
    lib = _load("snappy")
    src = _np(data)
    lib.snappy_max_compressed_length.restype = ctypes.c_size_t
    lib.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]
    cap = lib.snappy_max_compressed_length(len(src))
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = ctypes.c_size_t(cap)
    lib.snappy_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    rc = lib.snappy_compress(src.ctypes.data, len(src), dst.ctypes.data, ctypes.byref(n))
    assert rc == 0
    return dst[:n.value].tobytes()


def snappy_decompress(data, cap):
    lib = _load("snappy")
    src = _np(data)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = ctypes.c_size_t(cap)
    lib.snappy_uncompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    rc = lib.snappy_uncompress(src.ctypes.data, len(src), dst.ctypes.data, ctypes.byref(n))
    if rc != 0:
        raise ValueError("snappy_uncompress failed: %d" % rc)
    return dst[:n.value].tobytes()


def zstd_compress(data, level=3):
    lib = _load("zstd")
    src = _np(data)
    lib.ZSTD_compressBound.restype = ctypes.c_size_t
    lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    cap = lib.ZSTD_compressBound(len(src))
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    lib.ZSTD_compress.restype = ctypes.c_size_t
    lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n = lib.ZSTD_compress(dst.ctypes.data, cap, src.ctypes.data, len(src), level)
    assert not lib.ZSTD_isError(ctypes.c_size_t(n))
    return dst[:n].tobytes()


def zstd_decompress(data, cap):
    lib = _load("zstd")
    src = _np(data)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    lib.ZSTD_decompress.restype = ctypes.c_size_t
    lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
    n = lib.ZSTD_decompress(dst.ctypes.data, cap, src.ctypes.data, len(src))
    if lib.ZSTD_isError(n):
        raise ValueError("ZSTD_decompress failed")
    return dst[:n].tobytes()
