#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (build container only).

  * copies the Zstd decoder fixtures the reference's own tests pin
    (src/test/resources/data/zstd, used by T/zstd/AbstractTestZstd.java:41-78,175-184);
  * cuts a small corpus sample (64 KiB slices of testdata/ files, T/benchmark/DataSet.java:28-89)
    into tests/golden/corpus_sample.bin + corpus_sample.json so GPU parity tests have real data
    (the GPU box has no /root/reference);
  * writes tests/golden/manifest.json: SHA-256 of the oracle's compressed output for every
    corpus file (whole file, single block) and every sample slice, per codec.  A JDK >= 22 box can
    diff these against the real Lz4JavaCompressor / SnappyJavaCompressor / ZstdJavaCompressor (tools/GoldenDump.java).
"""
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

SAMPLE = [  # (file, offset) -> 64 KiB slice
    ("calgary/book1", 0), ("calgary/book1", 393216), ("calgary/geo", 0), ("calgary/pic", 0), ("calgary/obj2", 65536),
    ("calgary/news", 131072), ("canterbury/kennedy.xls", 65536), ("canterbury/ptt5", 131072), ("canterbury/alice29.txt", 0),
    ("html", 0), ("urls.10K", 65536), ("geo.protodata", 0), ("house.jpg", 0), ("kppkn.gtb", 0), ("mapreduce-osdi-1.pdf", 0),
    ("artificial/aaa.txt", 0), ("artificial/alphabet.txt", 0), ("artificial/random.txt", 0), ("large/world192.txt", 1048576),
]


def main():
    from tests import oracle_lib
    o = oracle_lib.load()
    os.makedirs(os.path.join(GOLD, "zstd"), exist_ok=True)
    for name in ("with-checksum", "with-checksum.zst", "multiple-frames", "multiple-frames.zst", "offset-before-start.zst",
                 "bad-second-frame.zst", "incompressible", "large-rle"):
        shutil.copyfile(os.path.join(REF, "src/test/resources/data/zstd", name), os.path.join(GOLD, "zstd", name))

    blob = bytearray()
    index = []
    for rel, off in SAMPLE:
        data = open(os.path.join(REF, "testdata", rel), "rb").read()[off:off + 65536]
        index.append({"file": rel, "offset": off, "length": len(data), "blob_offset": len(blob)})
        blob += data
    open(os.path.join(GOLD, "corpus_sample.bin"), "wb").write(blob)

    def sha(b):
        return hashlib.sha256(b).hexdigest()

    for e in index:
        d = bytes(blob[e["blob_offset"]:e["blob_offset"] + e["length"]])
        e["sha256"] = sha(d)
        for codec in ("lz4", "snappy", "zstd"):
            c = o.compress(codec, d)
            e[codec] = {"compressed_length": len(c), "sha256": sha(c)}
    json.dump(index, open(os.path.join(GOLD, "corpus_sample.json"), "w"), indent=1)

    manifest = {}
    files = sorted(f for f in glob.glob(os.path.join(REF, "testdata", "**", "*"), recursive=True) if os.path.isfile(f))
    for f in files:
        rel = os.path.relpath(f, os.path.join(REF, "testdata"))
        d = open(f, "rb").read()
        entry = {"length": len(d), "sha256": sha(d)}
        for codec in ("lz4", "snappy", "zstd"):
            c = o.compress(codec, d)
            entry[codec] = {"compressed_length": len(c), "sha256": sha(c)}
        manifest[rel] = entry
    json.dump(manifest, open(os.path.join(GOLD, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("golden: %d sample slices (%d bytes), %d corpus files" % (len(index), len(blob), len(manifest)))
    full_corpus(o, sha)


# T/benchmark/DataSet.java:28-89 (silesia/* and large/E.coli are not in the checkout)
DATASET_ORDER = [
    "canterbury/alice29.txt", "canterbury/asyoulik.txt", "canterbury/cp.html", "canterbury/fields.c", "canterbury/grammar.lsp",
    "canterbury/kennedy.xls", "canterbury/lcet10.txt", "canterbury/plrabn12.txt", "canterbury/ptt5", "canterbury/sum", "canterbury/xargs.1",
    "calgary/bib", "calgary/book1", "calgary/book2", "calgary/geo", "calgary/news", "calgary/obj1", "calgary/obj2", "calgary/paper1",
    "calgary/paper2", "calgary/paper3", "calgary/paper4", "calgary/paper5", "calgary/paper6", "calgary/pic", "calgary/progc", "calgary/progl",
    "calgary/progp", "calgary/trans",
    "artificial/a.txt", "artificial/aaa.txt", "artificial/alphabet.txt", "artificial/random.txt", "artificial/uniform_ascii.bin",
    "large/bible.txt", "large/world192.txt",
    "geo.protodata", "house.jpg", "html", "kppkn.gtb", "mapreduce-osdi-1.pdf", "urls.10K",
]


def full_corpus(o, sha):
    """The whole test corpus as ONE xz blob + index (the GPU box has no /root/reference), and tests/golden/oracle_manifest.tsv:
    one line per (file, offset, length, codec) with the length and SHA-256 of the oracle's compressed stream -- whole files (one call)
    and the block cuts of BASELINE configs[4] (64 KiB for LZ4 / Snappy, 128 KiB for Zstd; the last partial block included).
    tools/GoldenDump.java writes the same lines from the real Java classes (-> tests/golden/java_manifest.tsv)."""
    import lzma
    blob = bytearray()
    index = []
    for rel in DATASET_ORDER:
        d = open(os.path.join(REF, "testdata", rel), "rb").read()
        index.append({"file": rel, "offset": len(blob), "length": len(d), "sha256": sha(d)})
        blob += d
    open(os.path.join(GOLD, "corpus_full.bin.xz"), "wb").write(lzma.compress(bytes(blob), preset=9 | lzma.PRESET_EXTREME))
    json.dump(index, open(os.path.join(GOLD, "corpus_full.json"), "w"), indent=1)
    lines = []
    for e in index:
        d = bytes(blob[e["offset"]:e["offset"] + e["length"]])
        for codec, cut in (("lz4", 65536), ("snappy", 65536), ("zstd", 131072)):
            c = o.compress(codec, d)
            lines.append("%s\t0\t%d\t%s\t%d\t%s" % (e["file"], len(d), codec, len(c), sha(c)))
            if len(d) > cut:
                for off in range(0, len(d), cut):
                    piece = d[off:off + cut]
                    c = o.compress(codec, piece)
                    lines.append("%s\t%d\t%d\t%s\t%d\t%s" % (e["file"], off, len(piece), codec, len(c), sha(c)))
    open(os.path.join(GOLD, "oracle_manifest.tsv"), "w").write("\n".join(lines) + "\n")
    print("golden: full corpus %d files, %d bytes; oracle manifest %d lines" % (len(index), len(blob), len(lines)))
    stream_manifest(o, sha, [(e["file"], bytes(blob[e["offset"]:e["offset"] + e["length"]])) for e in index])


def stream_manifest(o, sha, files):
    """tests/golden/oracle_stream_manifest.tsv: what the oracle's ZstdOutputStream restatement writes for every corpus file and for the
    whole corpus as one stream ("*"); tools/java/GoldenStreamDump.java writes the same lines from the real class."""
    lines = []
    for name, d in files + [("*", b"".join(d for _, d in files))]:
        c = o.zstd_stream_compress(d)
        lines.append("%s\t0\t%d\tzstdstream\t%d\t%s" % (name, len(d), len(c), sha(c)))
    open(os.path.join(GOLD, "oracle_stream_manifest.tsv"), "w").write("\n".join(lines) + "\n")
    print("golden: stream manifest %d lines" % len(lines))


if __name__ == "__main__":
    main()
