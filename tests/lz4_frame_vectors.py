"""The vectors of T/lz4/TestLz4FrameDecompressor.java:61-230, rebuilt: its FrameBuilder writes ONE frame holding CONTENT as a
single compressed block, with the optional fields / checksums of the FLG byte, and the tests tamper with single fields.
`cases(o)` yields (name, frame bytes, output capacity, expected plaintext or None, expected message fragment or None)."""
import struct

MAGIC = 0x184D2204
FLG_VERSION = 1 << 6
FLG_BLOCK_INDEPENDENCE = 1 << 5
FLG_BLOCK_CHECKSUM = 1 << 4
FLG_CONTENT_SIZE = 1 << 3
FLG_CONTENT_CHECKSUM = 1 << 2
FLG_DICTIONARY_ID = 1
FLG_RESERVED_MASK = 0x02
BD_4MB = 7 << 4

CONTENT = b"".join(b"aircompressor lz4 frame test content %d " % i for i in range(200))


def build(o, flg, bd=BD_4MB, content=CONTENT, content_size=None, block_checksum=None, content_checksum=None, stored=False):
    flg |= FLG_VERSION
    out = bytearray(struct.pack("<I", MAGIC))
    desc = bytearray([flg & 0xFF, bd & 0xFF])
    if flg & FLG_CONTENT_SIZE:
        desc += struct.pack("<q", len(content) if content_size is None else content_size)
    if flg & FLG_DICTIONARY_ID:
        desc += struct.pack("<I", 0)
    out += desc
    out.append((o.xxh32(bytes(desc)) >> 8) & 0xFF)
    block = content if stored else o.compress("lz4", content)
    out += struct.pack("<I", len(block) | (0x80000000 if stored else 0))
    out += block
    if flg & FLG_BLOCK_CHECKSUM:
        out += struct.pack("<I", o.xxh32(block) if block_checksum is None else block_checksum)
    out += struct.pack("<I", 0)
    if flg & FLG_CONTENT_CHECKSUM:
        out += struct.pack("<I", o.xxh32(content) if content_checksum is None else content_checksum)
    return bytes(out)


def skippable(payload, nibble=0):
    return struct.pack("<II", 0x184D2A50 | nibble, len(payload)) + payload


def cases(o):
    n = len(CONTENT)
    ind = FLG_BLOCK_INDEPENDENCE
    f = build(o, ind)
    yield "plain", f, n, CONTENT, None
    yield "content checksum", build(o, ind | FLG_CONTENT_CHECKSUM), n, CONTENT, None
    yield "bad content checksum", build(o, ind | FLG_CONTENT_CHECKSUM, content_checksum=o.xxh32(CONTENT) ^ 1), n, None, "invalid content checksum"
    yield "block checksum", build(o, ind | FLG_BLOCK_CHECKSUM), n, CONTENT, None
    yield "bad block checksum", build(o, ind | FLG_BLOCK_CHECKSUM, block_checksum=o.xxh32(CONTENT) ^ 1), n, None, "invalid block checksum"
    yield "content size", build(o, ind | FLG_CONTENT_SIZE), n, CONTENT, None
    yield "bad content size", build(o, ind | FLG_CONTENT_SIZE, content_size=n + 1), n, None, "content size does not match"
    yield "linked blocks", build(o, 0), n, None, "linked blocks are not supported"
    yield "dictionary", build(o, ind | FLG_DICTIONARY_ID), n, None, "dictionary are not supported"
    yield "reserved flg", build(o, ind | FLG_RESERVED_MASK), n, None, "reserved bits"
    yield "reserved bd", build(o, ind, bd=0x71), n, None, "reserved bits"
    bad = bytearray(f)
    bad[0] ^= 0xFF
    yield "bad magic", bytes(bad), n, None, "magic number"
    bad = bytearray(f)
    bad[4] &= 0x3F
    yield "version", bytes(bad), n, None, "Unsupported LZ4 frame version"
    yield "block max size id", build(o, ind, bd=0x10), n, None, "block maximum size"
    bad = bytearray(f)
    bad[6] ^= 0xFF
    yield "header checksum", bytes(bad), n, None, "invalid header checksum"
    fc = build(o, ind | FLG_CONTENT_CHECKSUM)
    yield "three frames", fc * 3, 3 * n, CONTENT * 3, None
    sk = skippable(b"ignored metadata")
    yield "skippable frames", sk + f + sk + f + sk, 2 * n, CONTENT * 2, None
    yield "empty skippable", skippable(b"") + f, n, CONTENT, None
    yield "truncated skippable", f + sk[:-1], n, None, "Truncated LZ4 skippable frame"
    yield "trailing garbage", f + bytes([1, 2, 3, 4, 5]), n, None, "magic number"
    # beyond the reference's list: stored block, every block size id, all optional fields at once, short inputs, small outputs
    yield "stored block", build(o, ind | FLG_CONTENT_CHECKSUM, stored=True), n, CONTENT, None
    for bd_id in (4, 5, 6, 7):
        yield "bd %d" % bd_id, build(o, ind | FLG_BLOCK_CHECKSUM | FLG_CONTENT_SIZE | FLG_CONTENT_CHECKSUM, bd=bd_id << 4), n, CONTENT, None
    yield "too short", f[:6], n, None, "too short"
    yield "truncated header", f[:7] + b"", n, None, "missing block size"
    yield "output one short", f, n - 1, None, None
    yield "stored output short", build(o, ind, stored=True), n - 1, None, "Output buffer too small"
    yield "block past end", f[:len(f) - 10], n, None, "block extends past end"
    yield "missing end mark", f[:-4], n, None, "missing block size"
    yield "missing content checksum", fc[:-2], n, None, "missing content checksum"
    yield "skippable size missing", f + sk[:6], n, None, "missing frame size"
    yield "empty frame", build(o, ind, content=b"", stored=True)[:7] + struct.pack("<I", 0), 0, b"", None
