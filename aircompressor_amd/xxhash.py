"""Host-side mirror of the reference's one-shot hashing API (`io.airlift.compress.v3.xxhash`) over the C ABI.

`XxHash64HipHasher.hash(...)` / `XxHash32HipHasher.hash(...)` follow `XxHash64Hasher.hash(byte[] input, int offset,
int length, long seed)` (M/xxhash/XxHash64Hasher.java:55-86) and `XxHash32Hasher.hash(...)` (M/xxhash/XxHash32Hasher.java):
same argument order, same range check, and the result is the Java `long` / `int` (signed).  `hash_batch` hashes many
device-resident buffers per call.  HIP only: no CPU fallback.
"""
import ctypes

import numpy as np

from . import native
from .errors import IllegalArgumentException

DEFAULT_SEED = 0


def _check_from_index_size(data, offset, length):
    # Objects.checkFromIndexSize (M/xxhash/XxHash64JavaHasher.java:75, XxHash32JavaHasher.java:70)
    n = len(data)
    if offset < 0 or length < 0 or offset + length > n:
        raise IndexError("Range [%d, %d + %d) out of bounds for length %d" % (offset, offset, length, n))


class _HipHasher:
    _wide = True

    def __init__(self, device=0, native_ctx=None):
        self.native = native_ctx or native.HipNative(device)
        self._lib = self.native.lib

    def hash(self, input, offset=0, length=None, seed=DEFAULT_SEED):
        view = np.frombuffer(input, dtype=np.uint8)
        if length is None:
            length = view.size - offset
        _check_from_index_size(view, offset, length)
        src = view[offset:offset + length]
        ptr = src.ctypes.data if src.size else None
        if self._wide:
            out = ctypes.c_int64(0)
            r = self._lib.achip_xxhash64(self.native.ctx, ptr, int(src.size), ctypes.c_int64(_as_signed(seed, 64)), ctypes.byref(out))
        else:
            out = ctypes.c_int32(0)
            r = self._lib.achip_xxhash32(self.native.ctx, ptr, int(src.size), ctypes.c_int32(_as_signed(seed, 32)), ctypes.byref(out))
        if r < 0:
            native.raise_for_status(r)
        return int(out.value)

    def hash_batch(self, src_base, src_off, src_len, out_hash, n_buffers, seed=DEFAULT_SEED):
        """device pointers (ints or objects with data_ptr()); asynchronous on the context's stream"""
        p = lambda x: ctypes.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))  # noqa: E731
        if self._wide:
            r = self._lib.achip_xxhash64_batch(self.native.ctx, p(src_base), p(src_off), p(src_len), ctypes.c_int64(_as_signed(seed, 64)), p(out_hash), int(n_buffers))
        else:
            r = self._lib.achip_xxhash32_batch(self.native.ctx, p(src_base), p(src_off), p(src_len), ctypes.c_int32(_as_signed(seed, 32)), p(out_hash), int(n_buffers))
        if r < 0:
            native.raise_for_status(r)


def _as_signed(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


class XxHash64HipHasher(_HipHasher):
    """One-shot XXH64 (XxHash64Hasher.hash, M/xxhash/XxHash64Hasher.java:55-86); returns the Java long."""
    _wide = True


class XxHash32HipHasher(_HipHasher):
    """One-shot XXH32 (XxHash32Hasher.hash, M/xxhash/XxHash32Hasher.java); returns the Java int."""
    _wide = False


__all__ = ["XxHash64HipHasher", "XxHash32HipHasher", "DEFAULT_SEED", "IllegalArgumentException"]
