"""Oracle checks that need the reference checkout (build container only; skipped on the GPU box):
corpus round trips, cross-decoding with the third-party codecs the reference bundles, the
decoder's literal tables, and the stability of tests/golden/manifest.json."""
import glob
import hashlib
import json
import os
import re

import pytest

from tests import common, conftest, oracle_lib

pytestmark = pytest.mark.skipif(not conftest.reference_available(), reason="needs /root/reference")
REF = conftest.REFERENCE


def corpus_files():
    return sorted(f for f in glob.glob(os.path.join(REF, "testdata", "**", "*"), recursive=True) if os.path.isfile(f))


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def test_manifest_is_current(o):
    manifest = json.load(open(os.path.join(common.GOLDEN, "manifest.json")))
    for f in corpus_files():
        rel = os.path.relpath(f, os.path.join(REF, "testdata"))
        d = open(f, "rb").read()
        assert manifest[rel]["sha256"] == hashlib.sha256(d).hexdigest()
        for codec in ("lz4", "snappy", "zstd"):
            c = o.compress(codec, d)
            assert manifest[rel][codec]["sha256"] == hashlib.sha256(c).hexdigest(), (rel, codec)
            assert o.decompress(codec, c, len(d)) == d


def test_cross_decode_with_bundled_native_codecs(o):
    from tests import native_libs as nl
    for f in corpus_files():
        d = open(f, "rb").read()
        assert nl.lz4_decompress(o.compress("lz4", d), len(d)) == d, f       # our stream, their decoder
        assert o.decompress("lz4", nl.lz4_compress(d), len(d)) == d, f        # their stream, our decoder
        assert nl.snappy_decompress(o.compress("snappy", d), len(d)) == d, f
        assert o.decompress("snappy", nl.snappy_compress(d), len(d)) == d, f
        assert nl.zstd_decompress(o.compress("zstd", d), len(d)) == d, f           # oracle's level-3 frames, libzstd 1.5.6 decoder
        for i in range(0, min(len(d), 3 * 131072), 131072):
            b = d[i:i + 131072]
            assert nl.zstd_decompress(o.compress("zstd", b), len(b)) == b


def test_zstd_decoder_on_libzstd_frames(o):
    # T/zstd/TestZstd.java:21-47 exercises the Java decoder on zstd-jni level-3 frames; libzstd 1.5.6 stands in
    from tests import native_libs as nl
    for f in corpus_files():
        d = open(f, "rb").read()
        for level in (1, 3, 7):
            assert o.decompress("zstd", nl.zstd_compress(d, level), len(d)) == d, (f, level)
        for i in range(0, min(len(d), 4 * 131072), 131072):
            b = d[i:i + 131072]
            assert o.decompress("zstd", nl.zstd_compress(b, 3), len(b)) == b


def _java_int_array(text, name):
    m = re.search(name + r"\s*=\s*new\s+\w+\[\]\s*\{(.*?)\};", text, flags=re.S)
    return [int(x, 16) if x.lower().startswith("0x") else int(x) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1))]


def test_snappy_op_table_matches_java_source(o):
    text = open(os.path.join(REF, "src/main/java/io/airlift/compress/v3/snappy/SnappyRawDecompressor.java")).read()
    table = _java_int_array(text, "opLookupTable")
    assert len(table) == 256

    def entry(op):  # the layout restated in oracle/snappy.c and the HIP kernel
        kind, hi = op & 3, op >> 2
        if kind == 0:
            return hi + 1 if hi < 60 else (((hi - 59) << 11) | 1)
        if kind == 1:
            return (1 << 11) | ((hi >> 3) << 8) | ((hi & 7) + 4)
        return ((2 if kind == 2 else 4) << 11) | (hi + 1)

    assert [entry(i) for i in range(256)] == table


def test_zstd_default_tables_match_java_source(o):
    """oracle/zstd_dec.c builds the predefined LL/OF/ML decoding tables from the RFC 8878 distributions with
    the reference's own table builder; the result must equal the literal tables in ZstdFrameDecompressor.java:85-113.
    Decoding a frame whose sequences use predefined (BASIC) tables exercises them: offset-before-start.zst uses
    all-BASIC tables and must fail exactly at the corrupted offset, and libzstd frames of tiny inputs use BASIC mode."""
    from tests import native_libs as nl
    for n in (20, 60, 200, 1000):
        d = (b"abcdefgh" * 200)[:n] + bytes(range(n % 50))
        assert o.decompress("zstd", nl.zstd_compress(d, 3), len(d)) == d
    # and literally: parse the three Java tables and compare with tables dumped through a BASIC-mode decode
    text = open(os.path.join(REF, "src/main/java/io/airlift/compress/v3/zstd/ZstdFrameDecompressor.java")).read()
    blocks = re.findall(r"new FiniteStateEntropy\.Table\(\s*(\d+),\s*new int\[\] \{(.*?)\},\s*new byte\[\] \{(.*?)\},\s*new byte\[\] \{(.*?)\}\)", text, flags=re.S)
    assert len(blocks) == 3
    import ctypes
    import subprocess
    import tempfile
    src = r'''
    #include <stdio.h>
    #include "%s/oracle/zstd_dec.c"
    int main(void) { zctx* c = calloc(1, sizeof(zctx)); build_defaults(c);
      const fse_table* t[3] = {&c->default_ll, &c->default_of, &c->default_ml};
      for (int k = 0; k < 3; k++) { int n = 1 << t[k]->log2_size; printf("%%d\n", t[k]->log2_size);
        for (int i = 0; i < n; i++) printf("%%d ", t[k]->new_state[i]); printf("\n");
        for (int i = 0; i < n; i++) printf("%%d ", t[k]->symbol[i]); printf("\n");
        for (int i = 0; i < n; i++) printf("%%d ", t[k]->number_of_bits[i]); printf("\n"); }
      return 0; }''' % common.HERE.rsplit("/tests", 1)[0]
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "dump.c"), "w").write(src)
        subprocess.run(["gcc", "-O1", "-w", "-o", os.path.join(td, "dump"), os.path.join(td, "dump.c"),
                        os.path.join(common.HERE, "..", "oracle", "xxhash64.c")], check=True)
        lines = subprocess.run([os.path.join(td, "dump")], check=True, capture_output=True, text=True).stdout.strip().split("\n")
    for k, (log, ns, sym, nb) in enumerate(blocks):
        nums = lambda s: [int(x) for x in re.findall(r"-?\d+", s)]  # noqa: E731
        assert int(lines[4 * k]) == int(log)
        assert nums(lines[4 * k + 1]) == nums(ns)
        assert nums(lines[4 * k + 2]) == nums(sym)
        assert nums(lines[4 * k + 3]) == nums(nb)


def test_random_generator_feeds_codecs(o):
    for ratio in (0.1, 0.25, 0.5, 0.75, 1.0):
        g = o.random_generator(ratio).tobytes()
        for k in (0, 5):
            block = g[(k * 65536) % 1048576:][:65536]
            for codec in ("lz4", "snappy"):
                assert o.decompress(codec, o.compress(codec, block), 65536) == block
