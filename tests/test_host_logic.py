"""Host-checkable pieces of the device code: headers that are plain C are compiled with gcc and compared with the
oracle's restatement of the Java tables (no GPU involved)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECK = r"""
#include <stdio.h>
#include "oracle/zstd_dec.c"                     /* static tables restated from ZstdFrameDecompressor.java:68-83 */
#include "aircompressor_amd/csrc/zstd_codes.h"
int main(void)
{
    int bad = 0;
    for (int c = 0; c < 36; c++) {
        int32_t b, n;
        achip_zstd_ll_code(c, &b, &n);
        bad += b != LITERALS_LENGTH_BASE[c] || n != LITERALS_LENGTH_BITS[c];
    }
    for (int c = 0; c < 53; c++) {
        int32_t b, n;
        achip_zstd_ml_code(c, &b, &n);
        bad += b != MATCH_LENGTH_BASE[c] || n != MATCH_LENGTH_BITS[c];
    }
    for (int c = 0; c < 29; c++) {
        bad += achip_zstd_of_base(c) != OFFSET_CODES_BASE[c];
    }
    printf("%d\n", bad);
    return bad != 0;
}
"""


def test_zstd_code_arithmetic_equals_the_java_tables():
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "check.c")
        exe = os.path.join(tmp, "check")
        with open(src, "w") as f:
            f.write(CHECK)
        subprocess.run(["gcc", "-O1", "-std=gnu11", "-I", ROOT, "-I", os.path.join(ROOT, "oracle"), "-o", exe, src, os.path.join(ROOT, "oracle", "xxhash64.c")], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
        assert out.strip() == "0"


def test_sequence_stage_code_table_equals_the_java_tables():
    """zstd_decompress_pipe.hip's constant `seq_code_table` (round 4: the sequence stage's code -> baseline | extra bits << 24 table, literal-length codes at
    0.., match-length codes at 64..) parsed from the source text and compared, entry by entry, with the oracle's restatement of the Java tables
    (ZstdFrameDecompressor.java:68-83) -- printed by a small C program, as in the test above"""
    import re
    src = open(os.path.join(ROOT, "aircompressor_amd", "csrc", "zstd_decompress_pipe.hip")).read()
    body = src[src.index("__device__ const uint32_t seq_code_table[128] = {"):]
    body = body[body.index("{") + 1:body.index("};")]
    entries = []
    for tok in re.findall(r"ZC\((\d+), (\d+)\)|(?<![\w(])(0)(?=,)", body):
        entries.append((int(tok[0]), int(tok[1])) if tok[0] else (0, 0))
    assert len(entries) == 128, len(entries)
    prog = r"""
#include <stdio.h>
#include "oracle/zstd_dec.c"
int main(void)
{
    for (int c = 0; c < 36; c++) printf("%d %d\n", LITERALS_LENGTH_BASE[c], LITERALS_LENGTH_BITS[c]);
    for (int c = 0; c < 53; c++) printf("%d %d\n", MATCH_LENGTH_BASE[c], MATCH_LENGTH_BITS[c]);
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as tmp:
        cfile, exe = os.path.join(tmp, "t.c"), os.path.join(tmp, "t")
        open(cfile, "w").write(prog)
        subprocess.run(["gcc", "-O1", "-std=gnu11", "-I", ROOT, "-I", os.path.join(ROOT, "oracle"), "-o", exe, cfile, os.path.join(ROOT, "oracle", "xxhash64.c")], check=True)
        java = [tuple(int(x) for x in l.split()) for l in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n") if l]
    assert entries[:36] == java[:36] and entries[64:64 + 53] == java[36:]
    assert all(e == (0, 0) for e in entries[36:64] + entries[117:])


def test_decoder_choice_rule_on_the_cpu():
    """achip_device.h lz4_pick -- what auto mode's probes amount to -- compiled for the host (tools/hostemu's shim) and asked directly: the pooled
    bytes-per-sequence rule where no per-block count exists (the stream readers), the per-block rule where it does (round 4: a third of the sampled
    blocks short => two passes, whatever the pooled mean says), mixed groups, and "no record scratch => rings"."""
    import shutil
    import pytest
    clang = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ for the host build of the kernel source")
    prog = r"""
#include "hip/hip_runtime.h"
#include "achip_device.h"
#include <cstdio>
int main()
{
    struct C { int32_t s[8]; int32_t n, lim; };
    const C cases[] = {
        {{0, 100, 1100, 1, 0, 0}, 1024, 12},           // pooled: 44 bytes per sequence (units of 4: 11 < 12) -> two passes
        {{0, 100, 1300, 1, 0, 0}, 1024, 12},           // pooled: 52 -> rings
        {{0, 100, 1000000, 1, 400, 1024}, 1024, 12},   // per block: 400 of 1024 short (> a third) although the pooled mean is long -> two passes
        {{0, 100, 1000000, 1, 300, 1024}, 1024, 12},   // 300 of 1024 -> rings
        {{0, 100, 100, 1, 100, 1024}, 1024, 12},       // a tenth of the blocks short although the pooled mean is short -> rings
        {{0, 100, 100, 0, 1024, 1024}, 1024, 12},      // no record scratch -> rings, always
        {{17, 0, 0, 1, 0, 1024}, 1024, 12},            // more than a quarter of the 64 groups mixed -> two passes
        {{16, 0, 0, 1, 0, 1024}, 1024, 12},            // exactly a quarter -> rings
        {{0, 100, 500, 1, 0, 0}, 1024, 6},             // Snappy's limit (elements): 20 bytes per element -> two passes
        {{0, 100, 700, 1, 0, 0}, 1024, 6},             // 28 -> rings
    };
    for (const C& c : cases) printf("%d\n", achip::lz4_pick(c.s, c.n, c.lim));
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as tmp:
        cfile, exe = os.path.join(tmp, "t.cpp"), os.path.join(tmp, "t")
        open(cfile, "w").write(prog)
        subprocess.run([clang, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "tools", "hostemu"), "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "aircompressor_amd", "csrc"), "-o", exe, cfile], check=True)
        got = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [3, 0, 3, 0, 0, 0, 3, 0, 3, 0], got


def test_lane_private_decoder_kernels_on_the_cpu():
    """The kernel sources compiled for the host and run under tools/hostemu (every thread a fiber, cross-lane operations as rendezvous, lanes
    in a different order from pass to pass): the ring decoders with one lane per block and -- the product's default, the headline's kernel --
    with FOUR lanes per block (the lanes of a group meet at the emulator-only lockstep points of achip_rings.h), the lane-per-block decoders
    with an LDS window, and the cooperative two-pass decoders (parse to records + a wavefront per block; also with an arena so small that blocks
    fall back) -- plaintext, status, error offset and guard bands against the oracle, on the cases of the GPU parity suite, without a GPU."""
    import shutil
    import sys
    import pytest
    clang = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ for the host build of the kernel source")
    emu_dir = os.path.join(ROOT, "tools", "hostemu")
    subprocess.run([clang, "-O1", "-std=c++17", "-fPIC", "-shared", "-I", emu_dir, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "aircompressor_amd", "csrc"),
                    "-o", os.path.join(emu_dir, "libemu.so"), os.path.join(emu_dir, "emu.cpp")], check=True)
    # the checks run side by side (separate processes; the library is read-only): check_v3.py in three parts by decoder family
    def start(script, *args):
        return subprocess.Popen([sys.executable, os.path.join(emu_dir, script), *args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
    jobs = {
        "rings-1-lane": start("check_v3.py", "--quick", "--ops", "16,17,12,13"),
        "two-pass-lz4": start("check_v3.py", "--quick", "--ops", "24,25,26,27"),  # (26 / 27, 36 / 37: parsed by a wavefront per block)
        "two-pass-snappy": start("check_v3.py", "--quick", "--ops", "34,35,36,37"),
        "rings-4-lanes": start("check_v3.py", "--quick", "--ops", "44,54"),
        # the executor for records of any length (the Zstd pipeline's execute stage): LZ4 blocks re-expressed as Zstd-style records + one
        # literal buffer, executed and compared with the plaintext; records that run outside their buffers must be refused
        "records": start("check_records.py"),
        # the Hadoop block-stream reader's variant 2 (walk, chunks through the two-pass decoders with an arena sized after the chunk count is
        # read back, fold): LZ4 and Snappy streams at three buffer sizes against the plaintext
        "hadoop": start("check_hadoop.py"),
        # the LZ4 frame reader's variant 1 (walk, the frames' blocks as one batch through the two-pass decoder, fold with stored blocks,
        # content size and content checksum): the Java writer's frames and hand-built ones of 64 / 256 KiB blocks against the plaintext
        "lz4frame": start("check_lz4frame.py"),
        # the x-snappy-framed reader's variant 2 (walk, chunks through the two-pass Snappy decoder, CRC-32C verification, fold)
        "snappyframed": start("check_snappyframed.py"),
    }
    expected = {"rings-1-lane": 4, "two-pass-lz4": 4, "two-pass-snappy": 4, "rings-4-lanes": 2, "records": 1, "hadoop": 6, "lz4frame": 2, "snappyframed": 1}
    for name, job in jobs.items():
        out = job.communicate()[0]
        assert job.returncode == 0, (name, out)
        lines = [l for l in out.splitlines() if "mismatches" in l]
        assert len(lines) == expected[name] and all(l.endswith(" 0 mismatches") for l in lines), (name, out)


def test_zstd_pipeline_kernels_on_the_cpu():
    """The Zstd decode pipeline (parse / literals / sequences / execute / checksum and the multi-block stages: walk, per-block parse, tables
    fetched through links, repeat-offset sentinels, one window per frame) under tools/hostemu with quad-level rendezvous: frames of the
    oracle's encoder and of libzstd, single-block and multi-block, at two pass sizes, must decode to the plaintext on the fast path; damaged
    multi-block frames must either leave the fast path or decode to exactly what the oracle's decoder returns."""
    import shutil
    import sys
    import pytest
    clang = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ for the host build of the kernel source")
    emu_dir = os.path.join(ROOT, "tools", "hostemu")
    subprocess.run([clang, "-O1", "-std=c++17", "-fPIC", "-shared", "-I", emu_dir, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "aircompressor_amd", "csrc"),
                    "-o", os.path.join(emu_dir, "libemu_zstd.so"), os.path.join(emu_dir, "emu_zstd.cpp")], check=True)
    jobs = [subprocess.Popen([sys.executable, os.path.join(emu_dir, "check_zstd.py"), "--quick", "--part", part], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
            for part in ("single", "multi", "damaged")]  # (side by side)
    outs = [j.communicate()[0] for j in jobs]
    assert all(j.returncode == 0 for j in jobs), "\n".join(outs)
    out = "\n".join(outs)
    lines = [l for l in out.splitlines() if "mismatches" in l]
    assert len(lines) == 4 and all(" 0 mismatches" in l for l in lines), out
    assert "fast 18, fallback list []" in out and "fast 17, fallback list [17]" in out, out  # (the smallest passes: the frame of ~260 short blocks has no room)


def test_wavefront_per_item_kernels_on_the_cpu():
    """The wavefront-per-item kernels -- the LZ4 frame reader, the wavefront-per-stream readers of x-snappy-framed and Hadoop block streams,
    the one-kernel Zstd decoder (what every irregular item falls back to) -- under tools/hostemu (libemu_serial.so: wave_mem_order() and the
    rings' lockstep points are rendezvous there), driven by the GPU PARITY TESTS THEMSELVES through an emulator-backed harness
    (tools/hostemu/emu_harness.py): the reference's LZ4 frame vectors, every branch of the Hadoop readers, the framed format's error cases,
    the Zstd fixtures and corruptions -- output, status and error offset as the oracle has them.  (tools/hostemu/check_serial.py without
    --quick runs the long tests too: liblz4 / libzstd frames, random corruption of every container.)"""
    import shutil
    import sys
    import pytest
    clang = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ for the host build of the kernel source")
    emu_dir = os.path.join(ROOT, "tools", "hostemu")
    subprocess.run([clang, "-O1", "-std=c++17", "-fPIC", "-shared", "-I", emu_dir, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "aircompressor_amd", "csrc"),
                    "-o", os.path.join(emu_dir, "libemu_serial.so"), os.path.join(emu_dir, "emu_serial.cpp")], check=True)
    r = subprocess.run([sys.executable, os.path.join(emu_dir, "check_serial.py"), "--quick"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and r.stdout.strip().endswith("8 tests passed, 0 mismatches"), r.stdout


def test_whole_decode_paths_on_the_cpu():
    """Every decode kernel in ONE emulator library (tools/hostemu/libemu_all.so: the ring decoders, the two-pass decoders, the container
    readers, the one-kernel Zstd decoder and the Zstd pipeline with its multi-block stages; wave_mem_order() is a soft order point there), so
    that the product's own launch functions -- launch_hadoop_decompress, launch_lz4frame_decompress, launch_snappyframed_decompress,
    launch_zstd_decompress with every reader variant, the experimental ones included -- run start to end on the CPU, driven by the GPU
    parity tests themselves.  (tools/hostemu/check_serial.py --all without --quick runs the long tests too.)"""
    import shutil
    import sys
    import pytest
    clang = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ for the host build of the kernel source")
    emu_dir = os.path.join(ROOT, "tools", "hostemu")
    subprocess.run([clang, "-O1", "-std=c++17", "-fPIC", "-shared", "-I", emu_dir, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "aircompressor_amd", "csrc"),
                    "-o", os.path.join(emu_dir, "libemu_all.so"), os.path.join(emu_dir, "emu_all.cpp")], check=True)
    r = subprocess.run([sys.executable, os.path.join(emu_dir, "check_serial.py"), "--all", "--quick"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and r.stdout.strip().endswith("23 tests passed, 0 mismatches"), r.stdout


def test_encoders_and_writers_on_the_cpu():
    """The encoders -- LZ4, Snappy (one and two tiers), Zstd (match finder + entropy stage), the ZstdOutputStream writer, the LZ4 frame,
    x-snappy-framed and Hadoop block stream writers -- under tools/hostemu's access-granular lockstep (libemu_enc.so, built with clang's
    load / store tracing: every memory access of the kernel source is a soft order point, which is what replicated serial code with tables
    updated in place needs): bytes, lengths and statuses (capacities below the bound included) against the oracle's restatement of the Java
    encoders.  (tools/hostemu/check_enc.py without --quick: blocks beyond 64 / 128 / 256 KiB, every Zstd variant, the Hadoop buffer sizes;
    --chunked N: a ZstdOutputStream of N >= 4 MiB bytes through the chunked writer.)"""
    import shutil
    import sys
    import pytest
    clang = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ for the host build of the kernel source")
    emu_dir = os.path.join(ROOT, "tools", "hostemu")
    subprocess.run([clang, "-O2", "-std=c++17", "-fPIC", "-shared", "-fno-omit-frame-pointer", "-fsanitize-coverage=inline-8bit-counters,trace-loads,trace-stores", "-I", emu_dir,
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "aircompressor_amd", "csrc"),
                    "-o", os.path.join(emu_dir, "libemu_enc.so"), os.path.join(emu_dir, "emu_enc.cpp")], check=True)
    jobs = [subprocess.Popen([sys.executable, os.path.join(emu_dir, "check_enc.py"), "--quick", "--part", part], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
            for part in ("block", "zstd", "stream", "containers", "snappyfan")]  # (side by side)
    outs = [j.communicate()[0] for j in jobs]
    assert all(j.returncode == 0 for j in jobs), "\n".join(outs)
    assert all(o.strip().endswith("encoders under the emulator: 0 mismatches") for o in outs) and "MISMATCH" not in "".join(outs), "\n".join(outs)


def test_bench_java_random_generator_equals_the_oracles(oracle):
    """bench.py restates java.util.Random(301) + RandomGenerator in numpy (jump-ahead LCG) for its ratio sweep; the oracle's generator
    (oracle/misc.c, following T/snappy/RandomGenerator.java:25-74) is the checker."""
    import bench
    for ratio in (0.1, 0.25, 0.5, 0.75, 1.0, 0.001):
        assert (bench.java_random_generator(ratio) == oracle.random_generator(ratio)[:1048576]).all(), ratio


def test_bench_corpus_blocks_match_the_survey_counts():
    import bench
    assert bench.corpus_blocks(65536).size == 191 * 65536   # calgary 40 + canterbury 38 + large 97 + top-level 16
    assert bench.corpus_blocks(131072).size == 86 * 131072
