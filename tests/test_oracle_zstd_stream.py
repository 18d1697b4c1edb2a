"""The oracle's restatement of ZstdOutputStream (M/zstd/ZstdOutputStream.java:30-221; oracle/zstd_enc.c zstd_stream_compress): one
write(buffer, 0, n) + close(), as the reference's stream harness drives it (T/HadoopCodecCompressor.java:57-72).  No JVM here: the
restatement is pinned by what the Java source implies -- the frame compressor's bytes wherever the two must agree, the header forms,
the chunk arithmetic, valid frames for third-party decoders -- not by Java output (parity unpinned, like the encoders)."""
import numpy as np
import pytest

from tests import common, oracle_lib, native_libs


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def corpus(n):
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    return (whole * (n // len(whole) + 1))[:n]


def blocks_of(f):
    """[(type, size, sequences or None)] of the single frame f, and the frame header fields"""
    fhd = f[4]
    single = (fhd >> 5) & 1
    cs = fhd >> 6
    pos = 5
    window = None
    if not single:
        window = f[pos]
        pos += 1
    fcs_bytes = (1 if single else 0) if cs == 0 else 1 << cs
    fcs = int.from_bytes(f[pos:pos + fcs_bytes], "little") + (256 if cs == 1 else 0) if fcs_bytes else None
    pos += fcs_bytes
    out = []
    while True:
        h = int.from_bytes(f[pos:pos + 3], "little")
        pos += 3
        t, size = (h >> 1) & 3, h >> 3
        seqs = None
        if t == 2:
            bs = pos
            b0 = f[bs]
            lbt, sf = b0 & 3, (b0 >> 2) & 3
            if lbt < 2:
                hdr = 2 if sf == 1 else (3 if sf == 3 else 1)
                ls = (int.from_bytes(f[bs:bs + 2], "little") >> 4) if sf == 1 else ((int.from_bytes(f[bs:bs + 3], "little") >> 4) if sf == 3 else b0 >> 3)
                sect = hdr + (ls if lbt == 0 else 1)
            else:
                hdr = 3 if sf < 2 else (4 if sf == 2 else 5)
                hh = int.from_bytes(f[bs:bs + 5], "little")
                sect = hdr + ((hh >> 14) & 0x3FF if sf < 2 else ((hh >> 18) & 0x3FFF if sf == 2 else (hh >> 22) & 0x3FFFF))
            c = f[bs + sect]
            seqs = 0 if c == 0 else (c if c < 128 else (((c - 128) << 8) + f[bs + sect + 1] if c < 255 else int.from_bytes(f[bs + sect + 1:bs + sect + 3], "little") + 0x7F00))
        out.append((t, size, seqs))
        pos += 1 if t == 1 else size
        if h & 1:
            break
    assert pos + 4 == len(f)  # the checksum ends the frame
    return out, dict(fhd=fhd, single=single, window=window, fcs=fcs)


def test_below_four_mebibytes_the_stream_is_one_chunk_and_above_half_a_mebibyte_it_is_the_frame_compressors_output(o):
    """n < 4 MiB: close() writes everything as one chunk of known size.  The stream's parameters are the default row as it stands
    (window 2^20, hash 2^17, chain 2^16); ZstdFrameCompressor.compress adapts the row to the input size, which changes nothing once the
    input is beyond 512 KiB -- there the two must agree byte for byte."""
    for n in (524289, 700001, 1 << 20, (1 << 20) + 1, 3000000, (4 << 20) - 1):
        d = corpus(n)
        assert o.zstd_stream_compress(d) == o.compress("zstd", d), n
    for n in (0, 1, 100, 65536, 131072, 131073, 300000, 524288):
        d = corpus(n)
        s = o.zstd_stream_compress(d)
        blocks, h = blocks_of(s)
        assert h["single"] == 1 and h["fcs"] == n and h["fhd"] & 4      # window 1 MiB >= n: single segment, content size, checksum
        assert len(blocks) == max(1, (n + 131071) // 131072)
        assert o.decompress("zstd", s, n) == d
        if n > 0 and native_libs.available():
            assert native_libs.zstd_decompress(s, n) == d
    # single segment ends where the input outgrows the window
    _, h = blocks_of(o.zstd_stream_compress(corpus((1 << 20) + 1)))
    assert h["single"] == 0 and h["window"] == (20 - 10) << 3 and h["fcs"] == (1 << 20) + 1


@pytest.mark.parametrize("n", [4 << 20, (4 << 20) + 1, 7000000, 10 << 20])
def test_from_four_mebibytes_on_the_stream_flushes_slides_and_goes_blind_for_seven_blocks(o, n):
    """n >= 4 MiB: the first write fills the 4 MiB buffer, 23 blocks are flushed under a header WITHOUT a content size, the window slides by
    1920 KiB -- and BlockCompressionState.windowBaseOffset stays where enforceMaxDistance left it (1920 KiB), ahead of the moved data:
    the next 7 blocks (buffer positions 1024 .. 1920 KiB) cannot reference anything and carry no sequences.  From then on every 15
    blocks.  The frames are valid all the same."""
    d = corpus(n)
    s = o.zstd_stream_compress(d)
    blocks, h = blocks_of(s)
    assert h["single"] == 0 and h["fcs"] is None and h["window"] == (20 - 10) << 3 and h["fhd"] == 4
    assert len(blocks) == (n + 131071) // 131072
    slides = 1 + (n - (4 << 20)) // (1920 << 10)  # the buffer fills at 4 MiB and then with every further 1920 KiB; close() flushes without sliding
    blind = [b for j in range(slides) for b in range(23 + 15 * j, 23 + 15 * j + 7) if b < len(blocks)]
    for i, (t, size, seqs) in enumerate(blocks):
        full = i < len(blocks) - 1 or n % 131072 == 0
        if i in blind:
            assert t == 0 or seqs == 0, (i, t, seqs)
        elif full:
            assert t == 2 and seqs >= 1, (i, t, seqs)  # (the tiled corpus: long matches into the earlier copy, or thousands of short ones)
    assert o.decompress("zstd", s, n) == d
    if native_libs.available():
        assert native_libs.zstd_decompress(s, n) == d


def test_capacity_and_refusals(o):
    d = corpus(300000)
    s = o.zstd_stream_compress(d)
    assert o.zstd_stream_compress(d, cap=len(s)) == s
    with pytest.raises(oracle_lib.OracleError) as e:
        o.zstd_stream_compress(d, cap=len(s) - 1)
    assert e.value.cls == 2  # ACHIP_CLASS_OUTPUT_TOO_SMALL
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, 400000, dtype=np.uint8).tobytes()
    s = o.zstd_stream_compress(noise)
    blocks, _ = blocks_of(s)
    assert [t for t, _, _ in blocks] == [0, 0, 0, 0] and len(s) <= o.lib.orc_zstd_stream_max_compressed_length(len(noise))
    assert o.decompress("zstd", s, len(noise)) == noise
