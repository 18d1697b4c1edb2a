"""GPU test plumbing: torch owns the device buffers, the product C ABI does the work."""
import numpy as np


class GpuBatch:
    def __init__(self, device=0, options=None):
        import torch
        import aircompressor_amd as A
        self.torch = torch
        self.A = A
        self.dev = torch.device("cuda", device)
        self.codec = A.HipBatchCodec(device)
        for k, v in (options or {}).items():
            self.codec.native.set_option(k, v)

    def set_option(self, k, v):
        self.codec.native.set_option(k, v)

    def pack(self, blocks, align=16):
        lens = np.array([len(b) for b in blocks], dtype=np.int32)
        offs = np.zeros(len(blocks), dtype=np.int64)
        pos = 0
        for i, n in enumerate(lens):
            offs[i] = pos
            pos += (int(n) + align - 1) // align * align
        buf = np.zeros(max(pos, 16), dtype=np.uint8)
        for b, o in zip(blocks, offs):
            if len(b):
                buf[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
        return buf, offs, lens

    def run(self, op, blocks, caps, fill=0xA5, unaligned=False):
        """blocks: list[bytes]; caps: output capacity per block.  Returns (outputs, status, err_off).
        `op` may be a list with one op per block: the batch then goes through achip_mixed_batch (bucketed by codec)."""
        torch = self.torch
        n = len(blocks)
        src, src_off, src_len = self.pack(blocks, align=1 if unaligned else 16)
        caps = np.asarray(caps, dtype=np.int32)
        dst_off = np.zeros(n, dtype=np.int64)
        pos = 0
        for i, c in enumerate(caps):
            dst_off[i] = pos
            pos += int(c) if unaligned else (int(c) + 15) // 16 * 16
        d_src = torch.from_numpy(src).to(self.dev)
        d_dst = torch.full((max(pos, 16) + 64,), fill, dtype=torch.uint8, device=self.dev)  # +64 guard band
        d_src_off = torch.from_numpy(src_off).to(self.dev)
        d_src_len = torch.from_numpy(src_len).to(self.dev)
        d_dst_off = torch.from_numpy(dst_off).to(self.dev)
        d_dst_cap = torch.from_numpy(caps).to(self.dev)
        d_out_len = torch.full((n,), -7, dtype=torch.int32, device=self.dev)
        d_status = torch.full((n,), -7, dtype=torch.int32, device=self.dev)
        d_err = torch.zeros((n,), dtype=torch.int64, device=self.dev)
        torch.cuda.synchronize()
        if isinstance(op, (list, tuple, np.ndarray)):
            self.codec.launch_mixed(op, d_src, d_src_off, d_src_len, d_dst, d_dst_off, d_dst_cap, d_out_len, d_status, d_err, n)
        else:
            self.codec.launch(op, d_src, d_src_off, d_src_len, d_dst, d_dst_off, d_dst_cap, d_out_len, d_status, d_err, n)
        self.codec.synchronize()
        out = d_dst.cpu().numpy()
        out_len = d_out_len.cpu().numpy()
        status = d_status.cpu().numpy()
        err = d_err.cpu().numpy()
        # nothing may be written past a block's capacity: check the fill pattern between/after blocks
        ends = np.append(dst_off[1:], max(pos, 16)) if n else np.array([], dtype=np.int64)
        for i in range(n):
            gap = out[dst_off[i] + caps[i]:ends[i]]
            assert (gap == fill).all(), "block %d wrote past its capacity" % i
        assert (out[max(pos, 16):] == fill).all(), "wrote past the end of the destination buffer"
        outputs = [out[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() for i in range(n)]
        return outputs, status.tolist(), err.tolist()

    def run_host(self, op, blocks, caps, fill=0xA5):
        """The same through the host-pointer entry points (achip_batch_host / achip_mixed_batch_host): numpy buffers in and out."""
        n = len(blocks)
        src, src_off, src_len = self.pack(blocks, align=1)
        caps = np.asarray(caps, dtype=np.int32)
        dst_off = np.zeros(n, dtype=np.int64)
        pos = 0
        for i, c in enumerate(caps):
            dst_off[i] = pos
            pos += int(c) + 3
        dst = np.full(max(pos, 16) + 64, fill, dtype=np.uint8)
        if isinstance(op, (list, tuple, np.ndarray)):
            out_len, status, err = self.codec.run_host_mixed(op, src, src_off, src_len, dst, dst_off, caps)
        else:
            out_len, status, err = self.codec.run_host(op, src, src_off, src_len, dst, dst_off, caps)
        for i in range(n):
            assert (dst[dst_off[i] + caps[i]:dst_off[i] + caps[i] + 3] == fill).all(), "block %d wrote past its capacity" % i
        outputs = [dst[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() for i in range(n)]
        return outputs, status.tolist(), err.tolist()
