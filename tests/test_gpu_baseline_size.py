"""BASELINE.json's full-size configurations inside `pytest -m gpu`: until now they were only verified inside bench.py, so that a
bench regression and a parity regression were the same event.

* configs[1]: 262 144 x 64 KiB LZ4 blocks (and the same count of Snappy blocks) -- the corpus cut into 64 KiB blocks side by side with RandomGenerator-like
  synthetic blocks, every stream written by the ORACLE's encoder (= the Java encoder's bytes), decoded by the GPU's default path, every one of the 16 GiB
  of output bytes compared with the plaintext on the device; status 0 and the exact length for every block.
* configs[3]: 65 536 x 128 KiB Zstd frames as `ZstdFrameCompressor` writes them (the oracle's encoder), decoded by the five-stage pipeline, every byte compared,
  no item handed to the one-kernel decoder.

The pool of distinct blocks is small (what the oracle compresses in a second or two); the batch tiles it at distinct addresses, as bench.py does."""
import numpy as np
import pytest

from tests import common, oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def _pool(block_size, n_pool, seed):
    """n_pool distinct plaintext blocks of block_size bytes: every whole block of the corpus, then synthetic fragments blocks of five raw-fragment lengths"""
    blocks = []
    for name, data in sorted(common.corpus_full().items()):
        for at in range(0, len(data) - block_size + 1, block_size):
            blocks.append(bytes(data[at:at + block_size]))
    blocks = blocks[:n_pool * 3 // 4]
    rng = np.random.default_rng(seed)
    i = 0
    while len(blocks) < n_pool:
        raw = [10, 25, 50, 75, 100][i % 5]
        frags = rng.integers(0, 256, size=(block_size // 100 + 1, raw), dtype=np.uint8)
        blocks.append(np.tile(frags, (1, 100 // raw + 1))[:, :100].reshape(-1)[:block_size].tobytes())
        i += 1
    return blocks


def _run_tiled(op, comp, plain, block_size, reps, options=None):
    """decodes len(comp) x reps items (the pool tiled `reps` times at distinct addresses) and compares every output byte with the plaintext on the device"""
    import torch
    from tests.gpu_harness import GpuBatch
    g = GpuBatch(0, options=options)
    dev = g.dev
    k = len(comp)
    n = k * reps
    lens = np.array([len(c) for c in comp], dtype=np.int32)
    pad = (lens.astype(np.int64) + 15) // 16 * 16
    off = np.cumsum(pad) - pad
    tile_bytes = int(pad.sum())
    tile = np.zeros(tile_bytes, dtype=np.uint8)
    for o_, c in zip(off, comp):
        tile[o_:o_ + len(c)] = np.frombuffer(c, dtype=np.uint8)
    d_src = torch.from_numpy(tile).to(dev).repeat(reps)
    src_off = (np.arange(reps, dtype=np.int64)[:, None] * tile_bytes + off[None, :]).reshape(-1)
    src_len = np.tile(lens, reps)
    d_plain = torch.from_numpy(np.frombuffer(b"".join(plain), dtype=np.uint8).copy()).to(dev).view(k, block_size)
    d_dst = torch.full((n * block_size + 64,), 0xA5, dtype=torch.uint8, device=dev)
    a_so, a_sl = torch.from_numpy(src_off).to(dev), torch.from_numpy(src_len).to(dev)
    a_do = torch.arange(n, dtype=torch.int64, device=dev) * block_size
    a_dc = torch.full((n,), block_size, dtype=torch.int32, device=dev)
    o_len = torch.full((n,), -7, dtype=torch.int32, device=dev)
    st = torch.full((n,), -7, dtype=torch.int32, device=dev)
    eo = torch.zeros(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    g.codec.launch(op, d_src, a_so, a_sl, d_dst, a_do, a_dc, o_len, st, eo, n)
    g.codec.synchronize()
    assert int((st != 0).sum().item()) == 0, "statuses: %s" % st[st != 0][:8].tolist()
    assert int((o_len != block_size).sum().item()) == 0
    out = d_dst[:n * block_size].view(reps, k, block_size)
    step = max(1, (1 << 30) // (k * block_size))  # about a GiB per comparison
    for r in range(0, reps, step):
        chunk = out[r:r + step]
        assert bool((chunk == d_plain.unsqueeze(0)).all().item()), "output differs from the plaintext in repetitions %d .. %d" % (r, r + chunk.shape[0] - 1)
    assert bool((d_dst[n * block_size:] == 0xA5).all().item()), "wrote past the end of the destination buffer"
    return g


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_configs_1_262144_blocks_of_64_kib_oracle_streams(o, codec):
    import aircompressor_amd as A
    plain = _pool(65536, 512, 11)
    comp = [o.compress(codec, b) for b in plain]
    _run_tiled(A.OP_LZ4_DECOMPRESS if codec == "lz4" else A.OP_SNAPPY_DECOMPRESS, comp, plain, 65536, 512)


def test_configs_3_65536_java_encoder_frames_of_128_kib(o):
    import aircompressor_amd as A
    plain = _pool(131072, 128, 12)
    comp = [o.compress("zstd", b) for b in plain]
    g = _run_tiled(A.OP_ZSTD_DECOMPRESS, comp, plain, 131072, 512)
    assert g.codec.native.get_stat("zstd.decompress.fallback_items") == 0
