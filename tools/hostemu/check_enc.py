"""The ENCODERS and the stream / container WRITERS on the CPU (tools/hostemu/libemu_enc.so: every memory access of the kernel source is a soft
order point -- the lockstep of a wavefront at the granularity of memory operations, which is what code needs that runs the same serial
parse on all 64 lanes and updates its tables in place).  Output, status and length against the oracle's restatement of the Java encoders:
byte-identical or a mismatch.

  check_enc.py [--quick] [--part block|zstd|stream|containers]     the encoders' parity cases at sizes the emulator finishes in minutes
  check_enc.py --chunked N                                         one ZstdOutputStream of N bytes (N >= 4 MiB: the chunked writer with its
                                                                   window slides, behind zstd.stream.chunked = 1 in the product)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from emu_harness import EmuBatch, P
from tests import common, oracle_lib
from tests.oracle_lib import OracleError

o = oracle_lib.load()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu_enc.so"))
quick = "--quick" in sys.argv


class EncBatch(EmuBatch):
    def __init__(self, option=0, buffer_size=262144):
        self.lib = lib
        self.options = {}
        self.option = option
        self.buffer_size = buffer_size

    def _call(self, op, src, src_off, src_len, dst, dst_off, caps, out_len, status, err, n):
        return self.lib.emu_encode(op, P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(caps), P(out_len), P(status), P(err), n, self.option, self.buffer_size)


def expect(fn, *args):
    try:
        return fn(*args), 0
    except OracleError as e:
        return None, e.status


def compare(title, op, option, inputs, caps, reference, buffer_size=262144):
    t = time.time()
    outs, status, _ = EncBatch(option, buffer_size).run(op, inputs, caps)
    bad = 0
    for i, (b, c, s) in enumerate(zip(inputs, outs, status)):
        want, st = expect(reference, b, caps[i])
        if st != s or (st == 0 and c != want):
            bad += 1
            print("  MISMATCH %s: input %d (len %d, capacity %d): status %d / %d, %d bytes / %s" % (title, i, len(b), caps[i], s, st, len(c), "-" if want is None else len(want)))
    print("%-64s %4d items %8d bytes  %d mismatches  (%.0f s)" % (title, len(inputs), sum(len(b) for b in inputs), bad, time.time() - t), flush=True)
    return bad


def small_inputs(limit):
    rng = np.random.default_rng(7)
    blocks = [d for _, d in common.HAND_CASES if len(d) <= limit]
    sample = [d for _, d, _ in common.corpus_sample()]
    blocks += [d[:n] for d, n in zip(sample, (limit, limit // 2, 3000, 1000, limit, 777))]
    blocks += [b[:limit // 3] for b in common.synthetic_blocks(5, 3 if quick else 10)]
    blocks += [sample[0][:n] for n in ((1, 8, 13, 64, 255) if quick else range(1, 256, 11))]
    blocks += [bytes(2000), rng.integers(0, 256, 900, dtype=np.uint8).tobytes(), b"ab" * 700, rng.integers(0, 3, 2500, dtype=np.uint8).tobytes()]
    return blocks


def part_block():
    bad = 0
    blocks = small_inputs(6000 if quick else 20000)
    if not quick:
        blocks.append(common.corpus_sample()[1][1])  # a whole 64 KiB block: Snappy re-zeroes its table per 64 KiB, LZ4 uses the u16 table up to there
        blocks.append(b"".join(d for _, d, _ in common.corpus_sample()[:2])[:100000])  # beyond 64 KiB: the i32 table (LZ4), a second sub-block (Snappy)
    # (the serial-probe variants 0 run `if (lane == 0) { emit }` in the middle of replicated serial code and have the other lanes wait at the
    # join: what the shim's earliest-in-the-program-first release of paused lanes is for)
    # (LZ4 4 | 16 / 4 | 48: the window encoder as one wavefront per block / with two memory-tier wavefronts per workgroup; plain 4 is the default: one)
    for codec, op, variants in (("lz4", 1, (4, 4 | 16, 4 | 48, 1, 0)), ("snappy", 3, (4, 2, 1, 0))):  # (4: the window encoders, the defaults; Snappy 2 / 1 / 0: two tiers, one tier, serial probes)
        caps = [o.max_compressed_length(codec, len(b)) for b in blocks]
        for v in variants:
            bad += compare("%s compress, variant %d" % (codec, v), op, v, blocks, caps, lambda b, c, codec=codec: o.compress(codec, b, c))
        # capacities below the bound: what the Java encoder says ("output too small") or does, item by item
        some = blocks[2:8]
        tight = [max(len(o.compress(codec, b)) - k, 0) for k, b in enumerate(some)]
        bad += compare("%s compress, tight capacities" % codec, op, variants[0], some, tight, lambda b, c, codec=codec: o.compress(codec, b, c))
    return bad


def part_zstd():
    bad = 0
    sample = [d for _, d, _ in common.corpus_sample()]
    blocks = small_inputs(5000 if quick else 30000)
    blocks += [common.golden_zstd("large-rle")[:20000], common.golden_zstd("incompressible")[:9000]]
    if not quick:
        blocks.append((sample[0] + sample[1] + sample[2])[:140000])  # two blocks: tables, repeat offsets, Huffman reuse carried over
        blocks.append(b"".join(sample[:5])[:270000])                  # beyond 256 KiB: the default parameter row
    caps = [o.max_compressed_length("zstd", len(b)) for b in blocks]
    for v in ((3, 0, 1) if quick else (3, 0, 1, 2)):  # (3: match kernel with the window match finder + entropy kernel; 0: batch probes; 1: serial probes; 2: one kernel)
        bad += compare("zstd compress, variant %d" % v, 5, v, blocks, caps, lambda b, c: o.compress("zstd", b, c))
    some = blocks[2:8]
    tight = [max(len(o.compress("zstd", b)) - 3 * k, 0) for k, b in enumerate(some)]
    bad += compare("zstd compress, tight capacities", 5, 0, some, tight, lambda b, c: o.compress("zstd", b, c))
    return bad


def part_stream():
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(41)
    noise = rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()
    sizes = (0, 1, 7, 100, 4096, 16385) if quick else (0, 1, 7, 100, 4096, 16384, 16385, 100000, 131072, 131073, 262145)
    inputs = [whole[:n] for n in sizes] + [noise[:5000] if quick else noise, noise[:100] + whole[:3000] + noise[:3000], b"\0" * 50000, b"ab" * 20000]
    caps = [o.lib.orc_zstd_stream_max_compressed_length(len(b)) for b in inputs]
    bad = compare("ZstdOutputStream below 4 MiB (one chunk)", 14, 0, inputs, caps, lambda b, c: o.zstd_stream_compress(b, c))
    some = inputs[3:7]
    tight = [max(len(o.zstd_stream_compress(b)) - 1 - 5 * k, 0) for k, b in enumerate(some)]
    bad += compare("ZstdOutputStream, tight capacities", 14, 0, some, tight, lambda b, c: o.zstd_stream_compress(b, c))
    # capacities between the stream's size and the advertised bound: the Java stream succeeds (its blocks are compressed in its own buffer, the sink
    # only has to hold the result) -- every input at exactly its size, at one byte more, at 20 bytes more
    exact = [len(o.zstd_stream_compress(b)) + d for b in inputs for d in (0, 1, 20)]
    bad += compare("ZstdOutputStream, capacities of exactly the stream's size and a little more", 14, 0, [b for b in inputs for _ in range(3)], exact, lambda b, c: o.zstd_stream_compress(b, c))
    return bad


def part_containers():
    bad = 0
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, 9000, dtype=np.uint8).tobytes()
    inputs = [b"", b"x", whole[:700], whole[:5000], noise[:3000], whole[1000:3000] + noise[:2000] + whole[:1500], bytes(6000)]
    if not quick:
        inputs += [whole[:70000], whole[60000:140000], noise + whole[:20000]]
    for codec, op in (("lz4", 11), ("snappy", 13)):
        for buf in ((1024,) if quick else (1024, 70000, 262144)):
            caps = [o.hadoop_max_compressed_length(codec, len(b), buf) for b in inputs]
            bad += compare("Hadoop %s block stream writer, buffer %d" % (codec, buf), op, 0, inputs, caps, lambda b, c, codec=codec, buf=buf: o.hadoop_compress(codec, b, buf, c), buf)
    caps = [o.max_compressed_length("lz4frame", len(b)) for b in inputs]
    bad += compare("LZ4 frame writer", 7, 0, inputs, caps, lambda b, c: o.compress("lz4frame", b, c))
    caps = [o.max_compressed_length("snappyframed", len(b)) for b in inputs]
    for v in (1, 0):
        bad += compare("x-snappy-framed writer, variant %d" % v, 9, v, inputs, caps, lambda b, c: o.compress("snappyframed", b, c))
    return bad


def part_snappyfan():
    """Snappy buffers of several 64 KiB sub-blocks: the sub-blocks as work units of their own (snappy_compress.hip: list, encode into provisional places, fold),
    beside buffers of one sub-block, a capacity below the bound among them; option 4 | 16 is the same encoder with the sub-blocks in turn"""
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(19)
    noise = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    if quick:
        inputs = [whole[:65537], bytes(140000), whole[:3000], (b"abcdefgh" * 9000 + noise[:3000]) * 2, noise[:65536] + whole[:2], b"", whole[:131072 - 65000] + bytes(65000) + b"z"]
    else:
        inputs = [whole[:65537], whole[:131072], whole[5000:5000 + 200001], bytes(300000), whole[:3000], noise + whole[:70000] + noise[:1], b"", whole[:65536]]
    caps = [o.max_compressed_length("snappy", len(b)) for b in inputs]
    bad = 0
    for opt, what in ((4, "sub-blocks side by side"), (4 | 16, "sub-blocks in turn")):
        bad += compare("snappy compress, buffers beyond 64 KiB, %s" % what, 3, opt, inputs, caps, lambda b, c: o.compress("snappy", b, c))
    tight = [caps[0] - 1, caps[1], caps[2], caps[3] - 40]
    bad += compare("snappy compress, buffers beyond 64 KiB, capacities below the bound", 3, 4, inputs[:4], tight, lambda b, c: o.compress("snappy", b, c))
    return bad


def chunked(n):
    src = b"".join(common.multi_block_plains())
    while len(src) < n:
        src += src
    src = src[:n]
    want = o.zstd_stream_compress(src)
    t = time.time()
    outs, status, _ = EncBatch(1).run(14, [src], [len(want) + 4096])
    same = status[0] == 0 and outs[0] == want
    print("ZstdOutputStream of %d bytes (chunked writer): status %d, %d bytes against the oracle's %d: %s  (%.0f s)" % (n, status[0], len(outs[0]), len(want), "identical" if same else "MISMATCH", time.time() - t))
    if not same and status[0] == 0:
        k = next((i for i in range(min(len(want), len(outs[0]))) if want[i] != outs[0][i]), min(len(want), len(outs[0])))
        print("  first difference at byte %d" % k)
    return 0 if same else 1


def ostream(sizes):
    """the writer a chunk per launch (emu_zstd_ostream: the host schedule of achip_zstdstream_compress_feed / _finish over zstd_ostream_step_kernel)
    against the oracle's ZstdOutputStream: small streams in one closing step, a stream beyond 4 MiB with its flushes and buffer moves"""
    lib.emu_zstd_ostream.restype = ctypes.c_int64
    src_all = b"".join(common.multi_block_plains())
    while len(src_all) < max(sizes):
        src_all += src_all
    bad = 0
    for n in sizes:
        src = src_all[:n]
        want = o.zstd_stream_compress(src)
        for piece in ((1 << 30, 70001) if n < (1 << 20) else (1 << 20,)):
            t = time.time()
            out = np.zeros(len(want) + 4096, dtype=np.uint8)
            a = np.frombuffer(src, dtype=np.uint8) if n else np.zeros(1, dtype=np.uint8)
            r = lib.emu_zstd_ostream(P(a), ctypes.c_int64(n), P(out), ctypes.c_int64(len(out)), piece)
            same = r == len(want) and out[:r].tobytes() == want
            print("ZstdOutputStream of %d bytes, write() pieces of %d, a step per chunk: %d bytes against the oracle's %d: %s  (%.0f s)" % (n, piece, r, len(want), "identical" if same else "MISMATCH", time.time() - t), flush=True)
            bad += 0 if same else 1
    return bad


def main():
    if "--ostream" in sys.argv:
        k = sys.argv.index("--ostream")
        sizes = [int(x) for x in sys.argv[k + 1].split(",")] if len(sys.argv) > k + 1 else [0, 1, 100, 70000, 131072, 300000]
        sys.exit(1 if ostream(sizes) else 0)
    if "--chunked" in sys.argv:
        sys.exit(1 if chunked(int(sys.argv[sys.argv.index("--chunked") + 1])) else 0)
    parts = {"block": part_block, "zstd": part_zstd, "stream": part_stream, "containers": part_containers, "snappyfan": part_snappyfan}
    only = sys.argv[sys.argv.index("--part") + 1] if "--part" in sys.argv else None
    bad = 0
    for name, fn in parts.items():
        if only in (None, name):
            bad += fn()
    print("encoders under the emulator: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
