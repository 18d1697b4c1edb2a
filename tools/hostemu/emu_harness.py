"""An emulator-backed stand-in for tests/gpu_harness.GpuBatch: the same run(op, blocks, caps) -> (outputs, status, offsets), with the
wavefront-per-item kernels of tools/hostemu/libemu_serial.so doing the work on the CPU.  With it the GPU parity tests of the container
readers and of the Zstd decoder -- the reference's LZ4 frame vectors, every branch of the Hadoop readers, corruption with the oracle's
status and offset -- run without a GPU (tools/hostemu/check_serial.py), against the variant-0 kernels (a wavefront per item)."""
import ctypes, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


class _Native:
    def __init__(self, owner):
        self.owner = owner

    def set_option(self, k, v):
        self.owner.options[k] = v

    def get_stat(self, name):
        return -1


class _Codec:
    def __init__(self, owner):
        self.native = _Native(owner)


class NotEmulated(Exception):
    pass


class EmuBatch:
    variant = 0  # (tests/test_gpu_zstd.py: the one-kernel decoder)

    def __init__(self, options=None):
        self.lib = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu_serial.so"))
        self.options = dict(options or {})
        self.codec = _Codec(self)

    def set_option(self, k, v):
        self.options[k] = v

    def _call(self, op, src, src_off, src_len, dst, dst_off, caps, out_len, status, err, n):
        return self.lib.emu_serial(op, int(self.options.get("hadoop.buffer_size", 262144)), P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(caps), P(out_len), P(status), P(err), n)

    def run(self, op, blocks, caps, fill=0xA5, unaligned=False):
        n = len(blocks)
        caps = np.asarray(caps, dtype=np.int32)
        src_off = np.zeros(n, dtype=np.int64); src_len = np.zeros(n, dtype=np.int32); dst_off = np.zeros(n, dtype=np.int64)
        pos = 64
        for i, b in enumerate(blocks):
            src_off[i] = pos; src_len[i] = len(b); pos += len(b) if unaligned else (len(b) + 15) // 16 * 16
        src = np.full(pos + 64, 0x5A, dtype=np.uint8)
        for i, b in enumerate(blocks):
            if len(b):
                src[src_off[i]:src_off[i] + len(b)] = np.frombuffer(bytes(b), dtype=np.uint8)
        pos = 64
        for i, c in enumerate(caps):
            dst_off[i] = pos; pos += int(c) if unaligned else (int(c) + 15) // 16 * 16
        dst = np.full(pos + 64, fill, dtype=np.uint8)
        out_len = np.full(n, -7, dtype=np.int32); status = np.full(n, -7, dtype=np.int32); err = np.zeros(n, dtype=np.int64)
        if n:
            if isinstance(op, (list, tuple, np.ndarray)):
                raise NotEmulated("mixed batch")
            r = self._call(int(op), src, src_off, src_len, dst, dst_off, caps, out_len, status, err, n)
            if r != 0:
                raise NotEmulated("op %d" % op)
        ends = np.append(dst_off[1:], pos) if n else np.array([], dtype=np.int64)
        for i in range(n):
            assert (dst[dst_off[i] + caps[i]:ends[i]] == fill).all(), "block %d wrote past its capacity" % i
        assert (dst[:64] == fill).all() and (dst[pos:] == fill).all(), "wrote outside the destination buffer"
        return [dst[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() for i in range(n)], status.tolist(), err.tolist()


class EmuAllBatch(EmuBatch):
    """The same over libemu_all.so (every decode kernel, soft order points): whole product paths -- the default variants and the experimental
    ones, chosen by the same context options the GPU tests set."""

    def __init__(self, options=None):
        self.lib = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu_all.so"))
        self.options = dict(options or {})
        self.codec = _Codec(self)
        self.variant = self.options.get("zstd.decompress.variant", 1)
        self.counters = np.zeros(64, dtype=np.int32)
        self.codec.native.get_stat = self.get_stat

    def get_stat(self, name):
        words = {"zstd.decompress.fallback_items": 0, "zstd.decompress.multiblock_items": 40, "zstd.decompress.multiblock_blocks": 41, "zstd.decompress.multiblock_fast_items": 42}
        if name in words:
            return int(self.counters[words[name]])
        if name.startswith("zstd.decompress.fallback_stage"):
            return int(self.counters[32 + int(name[-1])])
        return -1

    def _call(self, op, src, src_off, src_len, dst, dst_off, caps, out_len, status, err, n):
        a = (P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(caps), P(out_len), P(status), P(err), n)
        o = self.options
        if op == 0:
            return self.lib.emu_batch({1: 44, 7: 24}.get(o.get("lz4.decompress.variant", 1), 44), *a)
        if op == 2:
            return self.lib.emu_batch({1: 54, 7: 34}.get(o.get("snappy.decompress.variant", 1), 54), *a)
        if op == 4:
            self.variant = o.get("zstd.decompress.variant", 1)
            return self.lib.emu_zstd_full(*a, int(self.variant), int(o.get("zstd.decompress.stream_blocks", 65536)), P(self.counters))
        if op == 6:
            return self.lib.emu_lz4frame(int(o.get("lz4frame.decompress.variant", 2)), *a)
        if op == 8:
            return self.lib.emu_snappyframed(int(o.get("snappyframed.decompress.variant", 3)), *a)
        if op in (10, 12):
            return self.lib.emu_hadoop(0, 1 if op == 12 else 0, int(o.get("hadoop.buffer_size", 262144)), int(o.get("hadoop.decompress.variant", 3)), *a)
        return -1
