"""Compress-side pin: the oracle's streams (tests/golden/oracle_manifest.tsv, written by tools/make_golden.py) against the REAL Java
encoders' streams (tests/golden/java_manifest.tsv, written by tools/java/run_golden_dump.sh on a box with a JDK >= 22 -- there is none
in the build container, so that file may be absent: the Java half is then skipped and compress-side parity stays "unpinned").

CPU suite: (1) the committed oracle manifest is what the oracle produces today on the shipped corpus (manifest not stale);
(2) if the Java manifest is there, it must equal the oracle manifest line for line.  The GPU suite (tests/test_gpu_corpus.py) compares the
GPU's streams with the same lines."""
import hashlib

import pytest

from tests import common


def test_oracle_manifest_is_current(oracle):
    rows = common.read_manifest_tsv("oracle_manifest.tsv")
    corpus = common.corpus_full()
    assert rows and len(rows) > 600
    files = set()
    for file, off, length, codec, clen, sha in rows:
        files.add(file)
        c = oracle.compress(codec, corpus[file][off:off + length])
        assert len(c) == clen and hashlib.sha256(c).hexdigest() == sha, (file, off, length, codec)
    assert files == set(corpus)  # every file of the corpus is pinned


def test_whole_file_lines_agree_with_round1_manifest():
    """tests/golden/manifest.json (round 1, generated from /root/reference directly) and the tsv describe the same whole-file streams"""
    import json
    import os
    m = json.load(open(os.path.join(common.GOLDEN, "manifest.json")))
    whole = {(f, codec): (clen, sha) for f, off, length, codec, clen, sha in common.read_manifest_tsv("oracle_manifest.tsv")
             if off == 0 and length == m[f]["length"]}
    for f, e in m.items():
        for codec in ("lz4", "snappy", "zstd"):
            assert whole[(f, codec)] == (e[codec]["compressed_length"], e[codec]["sha256"])


def test_java_manifest_equals_oracle_manifest():
    java = common.read_manifest_tsv("java_manifest.tsv")
    if java is None:
        pytest.skip("tests/golden/java_manifest.tsv absent: no JDK >= 22 has run tools/java/run_golden_dump.sh yet (compress-side parity unpinned)")
    orc = common.read_manifest_tsv("oracle_manifest.tsv")
    assert len(java) == len(orc)
    for j, o in zip(java, orc):
        assert j == o, "Java encoder and oracle differ: %r vs %r" % (j, o)


def test_oracle_stream_manifest_is_current(oracle):
    """tests/golden/oracle_stream_manifest.tsv (tools/make_golden.py): what the oracle's ZstdOutputStream restatement writes for every
    corpus file and for the whole corpus as one 14 MB stream ("*": chunks flushed before close(), window slides, blind blocks)."""
    rows = common.read_manifest_tsv("oracle_stream_manifest.tsv")
    corpus = common.corpus_full()
    assert len(rows) == len(corpus) + 1
    import json
    import os
    order = [e["file"] for e in json.load(open(os.path.join(common.GOLDEN, "corpus_full.json")))]
    for file, off, length, codec, clen, sha in rows:
        data = corpus[file] if file != "*" else b"".join(corpus[f] for f in order)
        assert off == 0 and length == len(data) and codec == "zstdstream"
        c = oracle.zstd_stream_compress(data)
        assert len(c) == clen and hashlib.sha256(c).hexdigest() == sha, file
        if file == "*":
            assert oracle.decompress("zstd", c, length) == data


def test_java_stream_manifest_equals_oracle_stream_manifest():
    java = common.read_manifest_tsv("java_stream_manifest.tsv")
    if java is None:
        pytest.skip("tests/golden/java_stream_manifest.tsv absent: no JDK >= 22 has run tools/java/run_golden_dump.sh yet (the stream writer's restatement is unpinned)")
    assert java == common.read_manifest_tsv("oracle_stream_manifest.tsv")

