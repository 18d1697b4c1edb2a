"""GPU parity on the WHOLE reference corpus (all 42 files, 14 MB: tests/golden/corpus_full.bin.xz) through the C ABI:

  * every file in one call per codec (LZ4 single block up to 4 MB, Snappy with its 64 KiB sub-blocks, Zstd multi-block frames):
    the GPU's stream must hash to the committed line of tests/golden/oracle_manifest.tsv (and of java_manifest.tsv when a JDK box
    has produced it), and the GPU decoders must restore the plaintext;
  * every 64 KiB (LZ4 / Snappy) and 128 KiB (Zstd) cut of every file, both directions, every decoder;
  * BASELINE configs[4]: ONE interleaved batch of (codec, direction) items over the corpus through the codec-bucketing scheduler
    (achip_mixed_batch, device-resident, and achip_mixed_batch_host), byte-exact per item.
Mirrors T/AbstractTestCompression.java:61-67 (testDecompress / testCompress over DataSet) for the Hip codecs."""
import hashlib
import threading

import numpy as np
import pytest

from tests import common, oracle_lib

pytestmark = pytest.mark.gpu

OPS = {"lz4": (1, 0), "snappy": (3, 2), "zstd": (5, 4)}  # codec -> (compress op, decompress op)
CUT = {"lz4": 65536, "snappy": 65536, "zstd": 131072}


@pytest.fixture(scope="module")
def gb():
    from tests.gpu_harness import GpuBatch
    return GpuBatch(0)


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def manifest_lines(codec, whole):
    """whole=True: the one-call-per-file lines; whole=False: the block cuts (a file that fits one cut is its own single cut)"""
    corpus = common.corpus_full()
    rows = []
    for r in common.read_manifest_tsv("oracle_manifest.tsv"):
        file, off, length = r[0], r[1], r[2]
        is_whole = off == 0 and length == len(corpus[file])
        if r[3] == codec and (is_whole if whole else (not is_whole or length <= CUT[codec])):
            rows.append(r)
    java = common.read_manifest_tsv("java_manifest.tsv")
    java = {(r[0], r[1], r[2], r[3]): r for r in java} if java else None
    return corpus, rows, java


def check_streams(codec, rows, outs, status, java):
    for (file, off, length, _, clen, sha), c, s in zip(rows, outs, status):
        assert s == 0, (file, off, length, s)
        assert len(c) == clen and hashlib.sha256(c).hexdigest() == sha, "%s %s@%d+%d: GPU stream differs from the oracle's" % (codec, file, off, length)
        if java is not None:
            assert java[(file, off, length, codec)][4:] == (clen, sha), "%s %s@%d+%d: GPU stream differs from the Java encoder's" % (codec, file, off, length)


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
def test_whole_files_one_call_each(gb, o, codec):
    corpus, rows, java = manifest_lines(codec, whole=True)
    assert len(rows) == 42
    plain = [corpus[r[0]] for r in rows]
    outs, status, _ = gb.run(OPS[codec][0], plain, [o.max_compressed_length(codec, len(b)) for b in plain])
    check_streams(codec, rows, outs, status, java)
    back, status, _ = gb.run(OPS[codec][1], outs, [len(b) for b in plain], unaligned=True)
    for r, b, p, s in zip(rows, plain, back, status):
        assert s == 0 and p == b, (codec, r[0], s)


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
def test_every_block_cut_both_directions(gb, o, codec):
    corpus, rows, java = manifest_lines(codec, whole=False)
    assert len(rows) > (100 if codec == "zstd" else 200)
    plain = [corpus[f][off:off + n] for f, off, n, *_ in rows]
    outs, status, _ = gb.run(OPS[codec][0], plain, [o.max_compressed_length(codec, len(b)) for b in plain])
    check_streams(codec, rows, outs, status, java)
    variants = {"lz4": [5, 1, 7], "snappy": [5, 1, 7], "zstd": [1, 0]}[codec]
    try:
        for v in variants:
            gb.set_option("%s.decompress.variant" % codec, v)
            back, status, _ = gb.run(OPS[codec][1], outs, [len(b) for b in plain], unaligned=(v % 2 == 0))
            for r, b, p, s in zip(rows, plain, back, status):
                assert s == 0 and p == b, (codec, v, r[0], r[1], s)
    finally:
        gb.set_option("%s.decompress.variant" % codec, variants[0])


def mixed_items(o, stride=1):
    """configs[4]: (op, input, capacity, expected output) for every whole file and every cut, all three codecs, both directions,
    interleaved deterministically (not grouped by codec)."""
    corpus = common.corpus_full()
    items = []
    rows = common.read_manifest_tsv("oracle_manifest.tsv")[::stride]
    for k, (f, off, n, codec, clen, sha) in enumerate(rows):
        p = corpus[f][off:off + n]
        c = o.compress(codec, p)
        assert hashlib.sha256(c).hexdigest() == sha
        cop, dop = OPS[codec]
        items.append((cop, p, o.max_compressed_length(codec, n), c))
        if n > 0:
            items.append((dop, c, n, p))
    order = np.random.default_rng(4).permutation(len(items))
    return [items[i] for i in order]


@pytest.mark.parametrize("concurrent", [1, 0], ids=["buckets-side-by-side", "buckets-in-turn"])
def test_mixed_batch_device_resident(gb, o, concurrent):
    """configs[4] as ONE call: the library buckets the items by codec op; by default every bucket runs on a helper context of its own (stream and
    scratch: mixed.concurrent = 1), twice in a row so that the second call meets the helpers of the first; or one bucket after the other"""
    items = mixed_items(o)
    assert len(items) > 1200 and len({it[0] for it in items}) == 6
    gb.set_option("mixed.concurrent", concurrent)
    try:
        for _ in range(2 if concurrent else 1):
            outs, status, _ = gb.run([it[0] for it in items], [it[1] for it in items], [it[2] for it in items])
            for k, (it, out, s) in enumerate(zip(items, outs, status)):
                assert s == 0, (k, it[0], len(it[1]), s)
                assert out == it[3], "item %d (op %d, %d bytes in)" % (k, it[0], len(it[1]))
    finally:
        gb.set_option("mixed.concurrent", 1)


def test_mixed_batch_host_pointers(gb, o):
    items = mixed_items(o, stride=2)
    gb.set_option("host.chunk_bytes", 4 << 20)  # many chunks: the staging pipeline wraps around its four slots many times
    try:
        outs, status, _ = gb.run_host([it[0] for it in items], [it[1] for it in items], [it[2] for it in items])
    finally:
        gb.set_option("host.chunk_bytes", 96 << 20)
    for k, (it, out, s) in enumerate(zip(items, outs, status)):
        assert s == 0, (k, it[0], len(it[1]), s)
        assert out == it[3], "item %d (op %d, %d bytes in)" % (k, it[0], len(it[1]))


def test_one_process_several_contexts(gb, o):
    """achip_multi_batch_host = java/.../HipBatchCodec.run executed: ONE process, N contexts, N host threads inside the library, contiguous
    byte-balanced slices; the mixed corpus batch (BASELINE configs[4]) and a homogeneous Zstd batch, byte-exact per item and identical to the
    single-context call.  N contexts on cuda:0 always (3 of them), and one context per device when the box has several."""
    import torch
    import aircompressor_amd as A
    items = mixed_items(o, stride=3)
    ops = [it[0] for it in items]
    blocks = [it[1] for it in items]
    caps = np.asarray([it[2] for it in items], dtype=np.int32)
    src, src_off, src_len = gb.pack(blocks, align=1)
    dst_off = np.concatenate([[0], np.cumsum(caps.astype(np.int64) + 5)[:-1]])
    single = np.full(int(dst_off[-1]) + int(caps[-1]) + 64, 0xA5, dtype=np.uint8)
    want_len, want_st, want_eo = gb.codec.run_host_mixed(ops, src, src_off, src_len, single, dst_off, caps)
    layouts = [[0, 0, 0]]
    if torch.cuda.device_count() > 1:
        layouts.append(list(range(torch.cuda.device_count())))
    for devices in layouts:
        multi = A.HipMultiContextCodec(devices=devices)
        try:
            for c in multi.contexts:
                c.set_option("host.chunk_bytes", 8 << 20)
            dst = np.full_like(single, 0xA5)
            out_len, st, eo = multi.run_host(ops, src, src_off, src_len, dst, dst_off, caps)
            starts = multi.slice_starts
            assert starts[0] == 0 and starts[-1] == len(items) and all(a < b for a, b in zip(starts, starts[1:])), starts
            assert (st == 0).all() and (out_len == want_len).all() and (st == want_st).all()
            assert (dst == single).all(), "several contexts wrote other bytes than one"
            for k, it in enumerate(items):
                assert dst[dst_off[k]:dst_off[k] + out_len[k]].tobytes() == it[3], (k, it[0])
            # homogeneous: every Zstd cut compressed, one damaged item keeps its status and the slices around it their results
            zrows = [r for r in common.read_manifest_tsv("oracle_manifest.tsv") if r[3] == "zstd"][:90]
            corpus = common.corpus_full()
            plain = [corpus[f][off:off + n] for f, off, n, *_ in zrows]
            comp = [o.compress("zstd", b) for b in plain]
            comp[40] = comp[40][:len(comp[40]) // 2]
            zsrc, zoff, zlen = gb.pack(comp, align=1)
            zcap = np.asarray([max(len(b), 1) for b in plain], dtype=np.int32)
            zdoff = np.concatenate([[0], np.cumsum(zcap.astype(np.int64))[:-1]])
            zdst = np.zeros(int(zcap.sum()) + 64, dtype=np.uint8)
            out_len, st, eo = multi.run_host(A.OP_ZSTD_DECOMPRESS, zsrc, zoff, zlen, zdst, zdoff, zcap)
            for k, b in enumerate(plain):
                if k == 40:
                    with pytest.raises(oracle_lib.OracleError) as e:
                        o.decompress("zstd", comp[k], len(b))
                    assert (st[k], eo[k]) == (e.value.status, e.value.offset)
                else:
                    assert st[k] == 0 and zdst[zdoff[k]:zdoff[k] + out_len[k]].tobytes() == b, k
        finally:
            multi.close()
    with pytest.raises(A.IllegalArgumentException):
        A.HipMultiContextCodec(contexts=[gb.codec.native, gb.codec.native]).run_host(A.OP_LZ4_COMPRESS, src, src_off, src_len, single, dst_off, caps)


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
def test_batch_host_hundreds_of_ragged_blocks(gb, o, codec):
    """achip_batch_host with n >> 1 (what HipBatchCodec.java calls): ragged real blocks, both directions, small and default chunking;
    malformed items keep their status without disturbing their neighbours."""
    corpus = common.corpus_full()
    rng = np.random.default_rng(12)
    blob = b"".join(corpus[f] for f in sorted(corpus))
    plain = []
    pos = 0
    while pos < len(blob) and len(plain) < 400:
        n = int(rng.choice([1, 13, 300, 4096, 20000, 65536, 100000])) if codec != "zstd" else int(rng.choice([1, 300, 4096, 65536, 131072]))
        plain.append(blob[pos:pos + n])
        pos += n
    cop, dop = OPS[codec]
    for chunk in (1 << 20, 192 << 20):
        gb.set_option("host.chunk_bytes", chunk)
        outs, status, _ = gb.run_host(cop, plain, [o.max_compressed_length(codec, len(b)) for b in plain])
        for b, c, s in zip(plain, outs, status):
            assert s == 0 and c == o.compress(codec, b)
        bad = list(outs)
        bad[7] = bad[7][:max(1, len(bad[7]) // 2)]
        bad[100] = b"\xff" * 40
        back, status, err = gb.run_host(dop, bad, [len(b) for b in plain])
        for k, (b, p, s) in enumerate(zip(plain, back, status)):
            if k in (7, 100):
                try:
                    o.decompress(codec, bad[k], len(b))
                    expect = (0, 0)
                except oracle_lib.OracleError as e:
                    expect = (e.status, e.offset)
                assert (s, err[k] if s else 0) == expect, (k, s, err[k], expect)
            else:
                assert s == 0 and p == b, (k, s)
    gb.set_option("host.chunk_bytes", 192 << 20)


def test_host_pipeline_options_change_no_byte(o):
    """The knobs of the host-pointer pipeline (round 6: host.slots, host.ramp -- smaller chunks at a batch's ends --, host.blit -- uploads / downloads by a copy kernel of
    the library's own --, host.copy_priority, host.look_max_blocks) over a batch of many chunks: outputs, statuses and error offsets are those of the default settings."""
    from tests.gpu_harness import GpuBatch
    import aircompressor_amd as A
    corpus = common.corpus_full()
    blob = b"".join(corpus[f] for f in sorted(corpus))
    rng = np.random.default_rng(21)
    plain = []
    pos = 0
    while pos < len(blob) and len(plain) < 500:
        n = int(rng.choice([7, 300, 4096, 20000, 65536, 70000]))
        plain.append(blob[pos:pos + n])
        pos += n
    comp = [o.compress("lz4", b) for b in plain]
    comp[11] = comp[11][:len(comp[11]) // 2]  # (a damaged item among them)
    caps = [len(b) for b in plain]
    want = None
    for options in ({}, {"host.slots": 2}, {"host.ramp": 0}, {"host.blit": 1, "host.blit_groups": 8}, {"host.blit": 2}, {"host.blit": 3, "host.slots": 3}, {"host.copy_priority": 0},
                    {"host.look_max_blocks": 4095}):
        g = GpuBatch(0, options=dict({"host.chunk_bytes": 1 << 20}, **options))
        got = g.run_host(A.OP_LZ4_DECOMPRESS, comp, caps)
        assert g.codec.native.get_stat("host.chunks") > 8
        if want is None:
            want = got
            assert all(s == 0 and p == b for k, (b, p, s) in enumerate(zip(plain, got[0], got[1])) if k != 11) and got[1][11] != 0
        else:
            assert got == want, options
        g.codec.native.close()


def test_two_contexts_two_threads(o):
    """include/aircompressor_hip.h: "distinct contexts may be used concurrently from distinct threads" (one Java codec object per thread,
    M/lz4/Lz4JavaCompressor.java:27-29; HipBatchCodec.java runs one thread per GPU): two contexts on device 0, two threads, each running
    compress + decompress of all three codecs on its own data at the same time; results equal the oracle's."""
    from tests.gpu_harness import GpuBatch
    corpus = common.corpus_full()
    data = [corpus["calgary/book1"], corpus["canterbury/ptt5"] + corpus["calgary/obj2"]]
    errors = []

    def worker(t):
        try:
            g = GpuBatch(0)
            for rounds in range(3):
                for codec in ("lz4", "snappy", "zstd"):
                    cut = CUT[codec]
                    plain = [data[t][i:i + cut] for i in range(0, len(data[t]), cut)]
                    cop, dop = OPS[codec]
                    outs, status, _ = g.run(cop, plain, [o.max_compressed_length(codec, len(b)) for b in plain])
                    assert all(s == 0 for s in status)
                    for b, c in zip(plain, outs):
                        assert c == o.compress(codec, b), (t, codec)
                    back, status, _ = g.run_host(dop, outs, [len(b) for b in plain])
                    assert all(s == 0 for s in status) and back == plain, (t, codec)
        except BaseException as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
