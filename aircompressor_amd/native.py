"""ctypes binding of libaircompressor_hip.so -- the Python twin of the Java `HipNative` FFM record
(java/io/airlift/compress/v3/hip/HipNative.java), modelled on M/lz4/Lz4Native.java:30-129.
"""
import ctypes
import os

from .errors import HipUnavailableError, IllegalArgumentException, MalformedInputException

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, "libaircompressor_hip.so")

CLASS_MALFORMED, CLASS_OUTPUT_TOO_SMALL, CLASS_INVALID_ARGUMENT, CLASS_DEVICE = 1, 2, 3, 4
DETAIL_LZ4_EMPTY_OUTPUT = 7

_i32, _i64, _vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
_BATCH = [_vp] * 10 + [_i32]

# name -> (restype, argtypes): every symbol include/aircompressor_hip.h declares
SIGNATURES = {
    "achip_status_class": (_i32, [_i32]),
    "achip_status_detail": (_i32, [_i32]),
    "achip_detail_message": (ctypes.c_char_p, [_i32]),
    "achip_version": (ctypes.c_char_p, []),
    "achip_device_count": (_i32, []),
    "achip_last_error": (ctypes.c_char_p, []),
    "achip_lz4_max_compressed_length": (_i32, [_i32]),
    "achip_snappy_max_compressed_length": (_i32, [_i32]),
    "achip_zstd_max_compressed_length": (_i32, [_i32]),
    "achip_snappy_uncompressed_length": (_i64, [_vp, _i64, ctypes.POINTER(_i64)]),
    "achip_zstd_decompressed_size": (_i64, [_vp, _i64, ctypes.POINTER(_i64)]),
    "achip_zstd_decompress_bound": (_i64, [_vp, _i64, ctypes.POINTER(_i64)]),
    "achip_ctx_create": (_vp, [_i32]),
    "achip_ctx_destroy": (None, [_vp]),
    "achip_ctx_device": (_i32, [_vp]),
    "achip_ctx_stream": (_vp, [_vp]),
    "achip_ctx_synchronize": (_i32, [_vp]),
    "achip_ctx_set_option": (_i32, [_vp, ctypes.c_char_p, _i64]),
    "achip_ctx_get_stat": (_i64, [_vp, ctypes.c_char_p]),
    "achip_snappyframed_max_compressed_length": (_i32, [_i32]),
    "achip_snappyframed_decompress_batch": (_i32, _BATCH),
    "achip_snappyframed_compress_batch": (_i32, _BATCH),
    "achip_snappyframed_compress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_snappyframed_decompress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_hadoop_max_compressed_length": (_i32, [_i32, _i32, _i32]),
    "achip_zstdstream_max_compressed_length": (_i32, [_i32]),
    "achip_zstdstream_compress_batch": (_i32, _BATCH),
    "achip_zstdstream_compress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_lz4hadoop_decompress_batch": (_i32, _BATCH),
    "achip_lz4hadoop_compress_batch": (_i32, _BATCH),
    "achip_snappyhadoop_decompress_batch": (_i32, _BATCH),
    "achip_snappyhadoop_compress_batch": (_i32, _BATCH),
    "achip_lz4hadoop_compress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_lz4hadoop_decompress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_snappyhadoop_compress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_snappyhadoop_decompress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_lz4frame_max_compressed_length": (_i32, [_i32]),
    "achip_lz4frame_decompress_batch": (_i32, _BATCH),
    "achip_lz4frame_compress_batch": (_i32, _BATCH),
    "achip_lz4frame_compress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_lz4frame_decompress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_xxhash64_batch": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i32]),
    "achip_xxhash32_batch": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _i32]),
    "achip_xxhash64": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "achip_xxhash32": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "achip_device_alloc": (_vp, [_vp, _i64]),
    "achip_device_free": (_i32, [_vp, _vp]),
    "achip_host_alloc_pinned": (_vp, [_i64]),
    "achip_host_free_pinned": (_i32, [_vp]),
    "achip_memcpy_h2d": (_i32, [_vp, _vp, _vp, _i64]),
    "achip_memcpy_d2h": (_i32, [_vp, _vp, _vp, _i64]),
    "achip_memset_d": (_i32, [_vp, _vp, _i32, _i64]),
    "achip_event_create": (_vp, []),
    "achip_event_destroy": (_i32, [_vp]),
    "achip_event_record": (_i32, [_vp, _vp]),
    "achip_event_elapsed_ms": (ctypes.c_float, [_vp, _vp]),
    "achip_lz4_decompress_batch": (_i32, _BATCH),
    "achip_lz4_compress_batch": (_i32, _BATCH),
    "achip_snappy_decompress_batch": (_i32, _BATCH),
    "achip_snappy_compress_batch": (_i32, _BATCH),
    "achip_zstd_decompress_batch": (_i32, _BATCH),
    "achip_zstd_compress_batch": (_i32, _BATCH),
    "achip_lz4_compress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_lz4_decompress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_snappy_compress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_snappy_decompress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_zstd_compress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_zstd_decompress": (_i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "achip_batch_host": (_i32, [_i32] + _BATCH),
    "achip_mixed_batch": (_i32, [_vp, _vp] + _BATCH[1:]),
    "achip_mixed_batch_host": (_i32, [_vp, _vp] + _BATCH[1:]),
    "achip_zstdstream_decompress_begin": (_vp, [_vp]),
    "achip_zstdstream_decompress_feed": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "achip_zstdstream_decompress_at_stopping_point": (_i32, [_vp]),
    "achip_zstdstream_decompress_end": (_i32, [_vp, _vp]),
    "achip_zstdstream_compress_begin": (_vp, [_vp]),
    "achip_zstdstream_compress_feed": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "achip_zstdstream_compress_finish": (_i32, [_vp, _vp, _vp, _i64, ctypes.POINTER(_i64)]),
    "achip_zstdstream_compress_end": (_i32, [_vp, _vp]),
    "achip_multi_batch_host": (_i32, [_vp, _i32, _i32, _vp] + _BATCH[1:] + [_vp]),
    "achip_partition_blocks": (_i32, [_vp, _i32, _i32, _vp]),
}

_lib = None


def load_library():
    """dlopen the in-tree shared library and type every symbol; raises HipUnavailableError if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBRARY_PATH):
            raise HipUnavailableError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C aircompressor_amd/csrc`); there is no CPU fallback" % LIBRARY_PATH)
        try:
            lib = ctypes.CDLL(LIBRARY_PATH)
        except OSError as e:  # e.g. libamdhip64 not loadable
            raise HipUnavailableError("cannot load %s: %s" % (LIBRARY_PATH, e))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = ABI/header drift: loud on purpose
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def status_class(status):
    return (-status) & 15 if status < 0 else 0


def status_detail(status):
    return (-status) >> 4 if status < 0 else 0


def raise_for_status(status, err_offset=0):
    """Translate a negative ACHIP status into the exception the Java codec would throw."""
    lib = load_library()
    cls, detail = status_class(status), status_detail(status)
    reason = lib.achip_detail_message(detail).decode()
    if cls == CLASS_MALFORMED:
        raise MalformedInputException(err_offset, reason, status)
    if cls == CLASS_DEVICE:
        raise HipUnavailableError("%s: %s" % (reason, lib.achip_last_error().decode()))
    raise IllegalArgumentException(reason, status)


class HipNative:
    """One HIP context (stream + scratch) on one device; owns its native handle."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.achip_device_count() <= 0:
            raise HipUnavailableError("no HIP device visible: the Hip codecs cannot run (no CPU fallback)")
        self.ctx = self.lib.achip_ctx_create(device)
        if not self.ctx:
            raise HipUnavailableError("achip_ctx_create(%d) failed: %s" % (device, self.lib.achip_last_error().decode()))
        self.device = device

    @staticmethod
    def is_enabled():
        try:
            return load_library().achip_device_count() > 0
        except HipUnavailableError:
            return False

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.achip_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        r = self.lib.achip_ctx_set_option(self.ctx, name.encode(), int(value))
        if r < 0:
            raise_for_status(r)

    def get_stat(self, name):
        return int(self.lib.achip_ctx_get_stat(self.ctx, name.encode()))

    def synchronize(self):
        r = self.lib.achip_ctx_synchronize(self.ctx)
        if r < 0:
            raise_for_status(r)

    @property
    def stream(self):
        return self.lib.achip_ctx_stream(self.ctx)
