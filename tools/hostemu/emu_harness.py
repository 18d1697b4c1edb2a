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
            r = self.lib.emu_serial(int(op), int(self.options.get("hadoop.buffer_size", 262144)), P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(caps), P(out_len), P(status), P(err), n)
            if r != 0:
                raise NotEmulated("op %d" % op)
        ends = np.append(dst_off[1:], pos) if n else np.array([], dtype=np.int64)
        for i in range(n):
            assert (dst[dst_off[i] + caps[i]:ends[i]] == fill).all(), "block %d wrote past its capacity" % i
        assert (dst[:64] == fill).all() and (dst[pos:] == fill).all(), "wrote outside the destination buffer"
        return [dst[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() for i in range(n)], status.tolist(), err.tolist()
