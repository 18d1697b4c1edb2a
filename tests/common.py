"""Shared test inputs: the reference harness' hand cases, the committed corpus sample, synthetic blocks."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

# T/AbstractTestCompression.java:47-56
HAND_CASES = [
    ("nothing", b""),
    ("short literal", b"hello world!"),
    ("small copy", b"XXXXabcdabcdABCDABCDwxyzwzyz123"),
    ("long copy", b"XXXXabcdefgh abcdefgh abcdefgh abcdefgh abcdefgh abcdefgh ABC"),
    ("long literal", bytes(range(256))),
]


def corpus_sample():
    """[(name, bytes, index_entry)] -- 64 KiB slices of the reference's corpora (tools/make_golden.py)."""
    blob = open(os.path.join(GOLDEN, "corpus_sample.bin"), "rb").read()
    index = json.load(open(os.path.join(GOLDEN, "corpus_sample.json")))
    return [("%s@%d" % (e["file"], e["offset"]), blob[e["blob_offset"]:e["blob_offset"] + e["length"]], e) for e in index]


_corpus_full = None


def corpus_full():
    """{file: bytes} -- every file of the reference's test corpus (T/benchmark/DataSet.java:28-89 as far as the checkout holds them:
    42 files, 14 MB), shipped as one xz blob + index by tools/make_golden.py so that it travels to the GPU box."""
    global _corpus_full
    if _corpus_full is None:
        import lzma
        blob = lzma.decompress(open(os.path.join(GOLDEN, "corpus_full.bin.xz"), "rb").read())
        index = json.load(open(os.path.join(GOLDEN, "corpus_full.json")))
        _corpus_full = {e["file"]: blob[e["offset"]:e["offset"] + e["length"]] for e in index}
    return _corpus_full


def read_manifest_tsv(name):
    """[(file, offset, length, codec, compressed_length, sha256)] of tests/golden/<name> (oracle_manifest.tsv / java_manifest.tsv)."""
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        return None
    rows = []
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if len(f) == 6:
            rows.append((f[0], int(f[1]), int(f[2]), f[3], int(f[4]), f[5]))
    return rows


def golden_zstd(name):
    return open(os.path.join(GOLDEN, "zstd", name), "rb").read()


def synthetic_blocks(seed, n_blocks, block_size=65536):
    """Mixed-statistics blocks: random-fragment data at several compressibilities (the shape of
    T/snappy/RandomGenerator.java), word-soup text, long runs, pure noise; ragged sizes at the end."""
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
    out = []
    for b in range(n_blocks):
        kind = b % 6
        if kind == 0:  # fragments: raw bytes repeated to 100
            ratio = [0.1, 0.25, 0.5, 0.75, 1.0][(b // 6) % 5]
            raw = max(1, int(100 * ratio))
            frags = rng.integers(0, 256, size=(block_size // 100 + 1, raw), dtype=np.uint8)
            data = np.tile(frags, (1, 100 // raw + 1))[:, :100].reshape(-1)[:block_size].tobytes()
        elif kind == 1:  # zipf word soup
            ids = np.minimum(rng.zipf(1.3, size=block_size // 3), len(words)) - 1
            data = b" ".join(words[i] for i in ids)[:block_size]
        elif kind == 2:  # runs of a single byte with noise
            data = bytearray()
            while len(data) < block_size:
                data += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 600))
                data += rng.integers(0, 256, size=int(rng.integers(0, 20)), dtype=np.uint8).tobytes()
            data = bytes(data[:block_size])
        elif kind == 3:  # noise
            data = rng.integers(0, 256, size=block_size, dtype=np.uint8).tobytes()
        elif kind == 4:  # short-period repeats (overlapping matches, offsets 1..40)
            data = bytearray()
            while len(data) < block_size:
                p = rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
                data += p * int(rng.integers(2, 60))
            data = bytes(data[:block_size])
        else:  # low-entropy bytes (few symbols)
            data = rng.integers(0, 4, size=block_size, dtype=np.uint8).tobytes()
        out.append(data)
    # ragged tail sizes
    for n in (0, 1, 2, 12, 13, 14, 64, 255, 4097, 65535):
        if n <= block_size:
            out.append(out[len(out) % max(1, n_blocks)][:n] if n_blocks else b"")
    return out


def multi_block_plains():
    """inputs beyond one block: corpus text with cross-block history, noise (raw blocks), long runs (RLE blocks), and data whose blocks look
    alike, so that the encoders reuse the previous block's Huffman table (treeless literals) and FSE tables (repeat mode)"""
    whole = b"".join(d for _, d, _ in corpus_sample())
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
    logs = "".join("2026-09-%02d %02d:%02d:%02d host%d GET /api/v1/items/%d?user=%d status=%d bytes=%d\n" % (
        rng.integers(1, 29), rng.integers(0, 24), rng.integers(0, 60), rng.integers(0, 60), rng.integers(0, 9), rng.integers(0, 100000), rng.integers(0, 5000),
        [200, 200, 200, 404, 500][rng.integers(0, 5)], rng.integers(100, 99999)) for _ in range(9000)).encode()
    # (libzstd level 3 cuts this into ~10 KiB blocks, raw ones among them: a frame of ~260 blocks)
    pool = [rng.integers(0, 256, int(rng.integers(20, 200)), dtype=np.uint8).tobytes() for _ in range(3000)]
    mixed = bytearray()
    while len(mixed) < (3 << 20):
        mixed += pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.5 else rng.integers(0, 256, int(rng.integers(20, 200)), dtype=np.uint8).tobytes()
    # sequences of one shape (literal length and match length always the same: libzstd level 19 describes such tables in RLE mode, and
    # repeats them), and blocks whose literals are one byte value (RLE literals, treeless literals, a block without sequences)
    prefix = rng.integers(0, 256, 65536, dtype=np.uint8)

    def fixed(n, mlen, lit):
        out = [prefix.tobytes()]
        size = len(prefix)
        while size < n:
            r = int(rng.integers(0, len(prefix) - mlen))
            out.append(prefix[r:r + mlen].tobytes())
            out.append(rng.integers(0, 256, lit, dtype=np.uint8).tobytes())
            size += mlen + lit
        return b"".join(out)[:n]

    def one_literal(n):
        base = bytes(rng.integers(97, 123, 150000, dtype=np.uint8).tolist())
        out = [base]
        size = len(base)
        while size < n:
            r = int(rng.integers(0, len(base) - 40))
            k = int(rng.integers(6, 30))
            out.append(base[r:r + k])
            out.append(b"z")
            size += k + 1
        return b"".join(out)[:n]
    shaped = [fixed(700000, 8, 1), fixed(700000, 16, 2), one_literal(800000)]
    return shaped + [whole, whole[:300000], whole[100000:100000 + 131073], whole[:131072] + noise[:140000] + whole[:70000], b"\0" * 400000, b"abc" * 100000 + whole[50000:250000],
            noise[:5], noise, whole[400000:1000000], b"q" * 131072 + b"r" * 131072 + whole[:10], whole[:262144], logs,
            (" ".join(str(x) for x in rng.integers(0, 1000, 200000))).encode(), bytes(rng.choice(list(b"ACGT"), 700000).tolist()), bytes(mixed[:3 << 20])]


def snappy_random_stream(rng, target):
    """A valid raw Snappy stream of about `target` plaintext bytes made of random elements of every kind (what the Java encoder never writes included: copies with
    4-byte offsets, runs behind runs, runs with one to three length bytes): for the parsers' rarely taken branches."""
    body, n = bytearray(), 0
    while n < target:
        kind = int(rng.integers(0, 10)) if n > 0 else 0
        if kind <= 2:  # a run, its length in the tag or in 1..3 bytes behind it
            ln = int(rng.choice([1, 2, 5, 16, 17, 31, 32, 33, 48, 60, 61, 64, 100, 256, 257, 1000, 70000][:15 if target < 100000 else 17]))
            data = bytes(rng.integers(97, 101, ln, dtype=np.uint8))
            if ln <= 60 and rng.integers(0, 4) != 0:
                body.append((ln - 1) << 2)
            else:
                nb = 1 if ln <= 256 else (2 if ln <= 65536 else 3)
                nb = min(3, nb + int(rng.integers(0, 2)))
                body.append((59 + nb) << 2)
                body += (ln - 1).to_bytes(nb, "little")
            body += data
            n += ln
        elif kind <= 5:  # a copy with a 1-byte offset: 4..11 bytes, offsets below 2048
            ln, off = int(rng.integers(4, 12)), int(rng.integers(1, min(n, 2047) + 1))
            body += bytes([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 0xFF])
            n += ln
        elif kind <= 8:  # a copy with a 2-byte offset: 1..64 bytes
            ln, off = int(rng.integers(1, 65)), int(rng.integers(1, min(n, 65535) + 1))
            body += bytes([2 | ((ln - 1) << 2)]) + off.to_bytes(2, "little")
            n += ln
        else:  # a copy with a 4-byte offset
            ln, off = int(rng.integers(1, 65)), int(rng.integers(1, n + 1))
            body += bytes([3 | ((ln - 1) << 2)]) + off.to_bytes(4, "little")
            n += ln
    pre, v = bytearray(), n
    while v >= 0x80:
        pre.append((v & 0x7F) | 0x80)
        v >>= 7
    pre.append(v)
    return bytes(pre + body), n
