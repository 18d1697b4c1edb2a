"""Batched, device-resident entry points (the hot path): thin wrapper over achip_*_batch.

Buffers are anything exposing `data_ptr()` (torch tensors on the HIP device) or raw integer
device addresses; no torch import here -- PyTorch is only the callers' allocator.
Work is sharded over GPUs one process per GPU by `partition_blocks` (contiguous, balanced
by bytes): blocks are independent, so there is no collective on the data path.
"""
import ctypes

import numpy as np

from . import native
from .native import HipNative

OP_LZ4_DECOMPRESS, OP_LZ4_COMPRESS, OP_SNAPPY_DECOMPRESS, OP_SNAPPY_COMPRESS, OP_ZSTD_DECOMPRESS, OP_ZSTD_COMPRESS, OP_LZ4FRAME_DECOMPRESS, OP_LZ4FRAME_COMPRESS, OP_SNAPPYFRAMED_DECOMPRESS, OP_SNAPPYFRAMED_COMPRESS, OP_LZ4HADOOP_DECOMPRESS, OP_LZ4HADOOP_COMPRESS, OP_SNAPPYHADOOP_DECOMPRESS, OP_SNAPPYHADOOP_COMPRESS, OP_ZSTDSTREAM_COMPRESS = range(15)
_FN = {
    OP_LZ4_DECOMPRESS: "achip_lz4_decompress_batch",
    OP_LZ4_COMPRESS: "achip_lz4_compress_batch",
    OP_SNAPPY_DECOMPRESS: "achip_snappy_decompress_batch",
    OP_SNAPPY_COMPRESS: "achip_snappy_compress_batch",
    OP_ZSTD_DECOMPRESS: "achip_zstd_decompress_batch",
    OP_ZSTD_COMPRESS: "achip_zstd_compress_batch",
    OP_LZ4FRAME_DECOMPRESS: "achip_lz4frame_decompress_batch",
    OP_LZ4FRAME_COMPRESS: "achip_lz4frame_compress_batch",
    OP_SNAPPYFRAMED_DECOMPRESS: "achip_snappyframed_decompress_batch",
    OP_SNAPPYFRAMED_COMPRESS: "achip_snappyframed_compress_batch",
    OP_LZ4HADOOP_DECOMPRESS: "achip_lz4hadoop_decompress_batch",
    OP_LZ4HADOOP_COMPRESS: "achip_lz4hadoop_compress_batch",
    OP_SNAPPYHADOOP_DECOMPRESS: "achip_snappyhadoop_decompress_batch",
    OP_SNAPPYHADOOP_COMPRESS: "achip_snappyhadoop_compress_batch",
    OP_ZSTDSTREAM_COMPRESS: "achip_zstdstream_compress_batch",
}


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


def partition_blocks(weights, n_parts):
    """Contiguous split of block indices balanced by `weights` (achip_partition_blocks)."""
    lib = native.load_library()
    w = np.ascontiguousarray(np.asarray(weights, dtype=np.int64))
    starts = np.zeros(n_parts + 1, dtype=np.int32)
    r = lib.achip_partition_blocks(w.ctypes.data, len(w), n_parts, starts.ctypes.data)
    if r < 0:
        native.raise_for_status(r)
    return starts


class HipMultiContextCodec:
    """ONE process, several contexts (normally one per device), one host thread per context inside the library: the Python twin of
    java/io/airlift/compress/v3/hip/HipBatchCodec.java's `run` (N contexts, N threads, byte-balanced contiguous slices) over
    achip_multi_batch_host.  Host numpy arrays in and out; units are independent, the slices exchange nothing."""

    def __init__(self, devices=None, contexts=None):
        self.lib = native.load_library()
        if contexts is None:
            if devices is None:
                devices = list(range(max(self.lib.achip_device_count(), 0)))
            contexts = [HipNative(d) for d in devices]
        if not contexts:
            raise native.HipUnavailableError("no HIP device visible: the Hip codecs cannot run (no CPU fallback)")
        self.contexts = list(contexts)
        self._handles = (ctypes.c_void_p * len(self.contexts))(*[c.ctx for c in self.contexts])
        self.slice_starts = None

    def run_host(self, op, src, src_off, src_len, dst, dst_off, dst_cap):
        """`op`: one OP_* for every item, or a sequence of one OP_* per item (a mixed batch, BASELINE configs[4])."""
        n = len(src_off)
        ops = None
        if not np.isscalar(op):
            ops = np.ascontiguousarray(op, dtype=np.int32)
            if len(ops) != n:
                raise native.IllegalArgumentException("ops must have one entry per item")
        src = np.ascontiguousarray(src, dtype=np.uint8)
        src_off = np.ascontiguousarray(src_off, dtype=np.int64)
        src_len = np.ascontiguousarray(src_len, dtype=np.int32)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.int64)
        dst_cap = np.ascontiguousarray(dst_cap, dtype=np.int32)
        out_len = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        err_off = np.zeros(n, dtype=np.int64)
        starts = np.zeros(len(self.contexts) + 1, dtype=np.int32)
        r = self.lib.achip_multi_batch_host(self._handles, len(self.contexts), 0 if ops is not None else int(op), ops.ctypes.data if ops is not None else None,
                                            src.ctypes.data, src_off.ctypes.data, src_len.ctypes.data, dst.ctypes.data, dst_off.ctypes.data, dst_cap.ctypes.data,
                                            out_len.ctypes.data, status.ctypes.data, err_off.ctypes.data, n, starts.ctypes.data)
        if r < 0:
            native.raise_for_status(r)
        self.slice_starts = starts
        return out_len, status, err_off

    def close(self):
        for c in self.contexts:
            c.close()


class HipBatchCodec:
    def __init__(self, device=0, native_ctx=None):
        self.native = native_ctx if native_ctx is not None else HipNative(device)
        self.lib = self.native.lib

    def launch(self, op, src, src_off, src_len, dst, dst_off, dst_cap, out_len, status, err_off, n_blocks):
        """Asynchronous on the context stream; all arguments device-accessible."""
        fn = getattr(self.lib, _FN[op])
        r = fn(self.native.ctx, _ptr(src), _ptr(src_off), _ptr(src_len), _ptr(dst), _ptr(dst_off), _ptr(dst_cap),
               _ptr(out_len), _ptr(status), _ptr(err_off), int(n_blocks))
        if r < 0:
            native.raise_for_status(r)

    def launch_mixed(self, ops, src, src_off, src_len, dst, dst_off, dst_cap, out_len, status, err_off, n_blocks):
        """A mixed batch (BASELINE configs[4]): item i is processed by ops[i] (OP_*), any interleaving.  `ops` is a HOST int32 array;
        everything else is device-accessible as for `launch`.  The items are bucketed by codec op inside the library (each kernel
        launch is homogeneous) and the results come back in item order.  Asynchronous on the context stream."""
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        if len(ops) != int(n_blocks):
            raise native.IllegalArgumentException("ops must have one entry per item")
        r = self.lib.achip_mixed_batch(self.native.ctx, ops.ctypes.data, _ptr(src), _ptr(src_off), _ptr(src_len), _ptr(dst), _ptr(dst_off), _ptr(dst_cap),
                                       _ptr(out_len), _ptr(status), _ptr(err_off), int(n_blocks))
        if r < 0:
            native.raise_for_status(r)

    def synchronize(self):
        self.native.synchronize()

    def run_host_mixed(self, ops, src, src_off, src_len, dst, dst_off, dst_cap):
        """Host numpy arrays in/out through achip_mixed_batch_host: one op per item."""
        n = len(src_off)
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        src = np.ascontiguousarray(src, dtype=np.uint8)
        src_off = np.ascontiguousarray(src_off, dtype=np.int64)
        src_len = np.ascontiguousarray(src_len, dtype=np.int32)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.int64)
        dst_cap = np.ascontiguousarray(dst_cap, dtype=np.int32)
        out_len = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        err_off = np.zeros(n, dtype=np.int64)
        r = self.lib.achip_mixed_batch_host(self.native.ctx, ops.ctypes.data, src.ctypes.data, src_off.ctypes.data, src_len.ctypes.data, dst.ctypes.data,
                                            dst_off.ctypes.data, dst_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data, err_off.ctypes.data, n)
        if r < 0:
            native.raise_for_status(r)
        return out_len, status, err_off

    def run_host(self, op, src, src_off, src_len, dst, dst_off, dst_cap):
        """Host numpy arrays in/out through achip_batch_host (stages through pinned memory)."""
        n = len(src_off)
        src = np.ascontiguousarray(src, dtype=np.uint8)
        src_off = np.ascontiguousarray(src_off, dtype=np.int64)
        src_len = np.ascontiguousarray(src_len, dtype=np.int32)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.int64)
        dst_cap = np.ascontiguousarray(dst_cap, dtype=np.int32)
        out_len = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        err_off = np.zeros(n, dtype=np.int64)
        r = self.lib.achip_batch_host(op, self.native.ctx, src.ctypes.data, src_off.ctypes.data, src_len.ctypes.data, dst.ctypes.data,
                                      dst_off.ctypes.data, dst_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data, err_off.ctypes.data, n)
        if r < 0:
            native.raise_for_status(r)
        return out_len, status, err_off

    # timing helpers on the context stream
    def event(self):
        return self.lib.achip_event_create()

    def record(self, ev):
        self.lib.achip_event_record(self.native.ctx, ev)

    def elapsed_ms(self, ev0, ev1):
        return float(self.lib.achip_event_elapsed_ms(ev0, ev1))
