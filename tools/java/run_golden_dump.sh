#!/usr/bin/env bash
# One command for whoever has a JDK >= 22: pins the encoders against the real Java classes.
#   tools/java/run_golden_dump.sh /path/to/aircompressor-checkout
# Compiles the reference's codec packages with plain javac (no Maven, no network: SURVEY.md Appendix C -- everything except
# the *Codec.java Hadoop adapters), runs tools/java/GoldenDump.java over <checkout>/testdata and writes
# tests/golden/java_manifest.tsv; then `python -m pytest tests/test_java_golden.py` (CPU) and `-m gpu` compare it with the
# oracle's manifest and with the GPU's streams.  Never run in the build container (no JVM there).
set -euo pipefail
REF=${1:?usage: run_golden_dump.sh <aircompressor checkout>}
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../.." && pwd)
M="$REF/src/main/java/io/airlift/compress/v3"
OUT=$(mktemp -d)
trap 'rm -rf "$OUT"' EXIT
find "$M" -maxdepth 1 -name '*.java' > "$OUT/sources.txt"
for pkg in internal lz4 snappy zstd xxhash; do
    find "$M/$pkg" -name '*.java' ! -name '*Codec.java' >> "$OUT/sources.txt"
done
for f in HadoopStreams HadoopInputStream HadoopOutputStream; do
    echo "$M/hadoop/$f.java" >> "$OUT/sources.txt"
done
echo "$HERE/GoldenDump.java" >> "$OUT/sources.txt"
echo "$HERE/GoldenStreamDump.java" >> "$OUT/sources.txt"
javac -d "$OUT/classes" @"$OUT/sources.txt"
java --enable-native-access=ALL-UNNAMED -cp "$OUT/classes" GoldenDump "$REF/testdata" > "$REPO/tests/golden/java_manifest.tsv"
wc -l "$REPO/tests/golden/java_manifest.tsv"
# ZstdOutputStream (every file, and the whole corpus as one stream of several chunks) against oracle_stream_manifest.tsv
java --enable-native-access=ALL-UNNAMED -cp "$OUT/classes" GoldenStreamDump "$REF/testdata" > "$REPO/tests/golden/java_stream_manifest.tsv"
if diff -q "$REPO/tests/golden/java_stream_manifest.tsv" "$REPO/tests/golden/oracle_stream_manifest.tsv" > /dev/null; then
    echo "STREAM WRITER PINNED: ZstdOutputStream and the oracle's restatement agree on every line"
else
    echo "STREAM WRITER MISMATCH: diff tests/golden/java_stream_manifest.tsv tests/golden/oracle_stream_manifest.tsv"
    diff "$REPO/tests/golden/java_stream_manifest.tsv" "$REPO/tests/golden/oracle_stream_manifest.tsv" | head -20
fi
if diff -q "$REPO/tests/golden/java_manifest.tsv" "$REPO/tests/golden/oracle_manifest.tsv" > /dev/null; then
    echo "PARITY PINNED: the Java encoders and the oracle produce identical streams for every line"
else
    echo "MISMATCH: diff tests/golden/java_manifest.tsv tests/golden/oracle_manifest.tsv"
    diff "$REPO/tests/golden/java_manifest.tsv" "$REPO/tests/golden/oracle_manifest.tsv" | head -20
    exit 1
fi
