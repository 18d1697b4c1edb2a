"""Runs GPU parity tests of tests/test_gpu_*.py on the CPU (--all: over libemu_all.so -- whole product paths, the default and the experimental
reader variants; otherwise over libemu_serial.so -- the wavefront-per-item kernels): their `gb` is an emulator-backed harness (tools/hostemu/emu_harness.py) whose
run() drives the wavefront-per-item kernels of libemu_serial.so -- the LZ4 frame reader, the wavefront-per-stream readers of x-snappy-framed
and Hadoop block streams, the one-kernel Zstd decoder.  Only tests that decode (and whose expectations come from the oracle) are run."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from emu_harness import EmuBatch, EmuAllBatch
from tests import oracle_lib

o = oracle_lib.load()


def main():
    import tests.test_gpu_lz4_frame as lf
    import tests.test_gpu_hadoop as hd
    import tests.test_gpu_zstd as zs
    import tests.test_gpu_snappy_framed as sf
    whole = "--all" in sys.argv
    # (--all: one run per reader variant, as the test modules' fixtures parametrise them -- the experimental ones included)
    variants = {"lz4 frame": [{"lz4frame.decompress.variant": v} for v in (2, 0, 1)], "hadoop streams": [{"hadoop.decompress.variant": v} for v in (3, 1, 0, 2)],
                "snappy framed": [{"snappyframed.decompress.variant": v} for v in (3, 1, 0, 2)], "zstd": [{"zstd.decompress.variant": v} for v in (1, 0)]} if whole else {}
    plan = [("lz4 frame", lf, [n for n in dir(lf) if n.startswith("test_")]),
            ("hadoop streams", hd, [n for n in dir(hd) if n.startswith("test_")]),
            ("snappy framed", sf, [n for n in dir(sf) if n.startswith("test_")]),
            ("zstd", zs, ["test_golden_fixtures", "test_error_fixtures_and_corruptions", "test_offsets_beyond_28_bits_are_rejected", "test_libzstd_frames_decode_to_plaintext",
                          "test_multi_block_frames_and_concatenated_frames", "test_damaged_multi_block_frames_report_what_the_oracle_reports"])]
    only = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else None
    quick = {"test_reference_vectors", "test_reader_branches", "test_reference_error_cases_and_chunk_kinds", "test_golden_fixtures", "test_error_fixtures_and_corruptions",
             "test_offsets_beyond_28_bits_are_rejected", "test_multi_block_frames_and_concatenated_frames"} if "--quick" in sys.argv else None
    bad = ran = 0
    for title, mod, names in plan:
        if only and only not in title:
            continue
        for name in names:
            if quick is not None and name not in quick:
                continue
            fn = getattr(mod, name)
            argnames = fn.__code__.co_varnames[:fn.__code__.co_argcount]
            if "gb" not in argnames and "gbd" not in argnames:
                continue  # (tests that build their own GPU context: twins, full-size properties)
            params = [{}]
            for mark in getattr(fn, "pytestmark", []):
                if mark.name == "parametrize":
                    key, values = mark.args[0], mark.args[1]
                    params = [dict(p, **{key: v}) for p in params for v in values]
            params = [dict(p, _options=opt) for p in params for opt in variants.get(title, [None])]
            for p in params:
                options = p.pop("_options")
                kwargs = dict(p)
                p = dict(p, **(options or {}))
                for a in argnames:
                    if a in ("gb", "gbd"):
                        kwargs[a] = EmuAllBatch(options) if whole else EmuBatch()
                    elif a == "o":
                        kwargs[a] = o
                if any(a not in kwargs for a in argnames):
                    continue
                t0 = time.time()
                try:
                    fn(**kwargs)
                    verdict = "ok"
                except AssertionError as e:
                    verdict = "FAILED: " + (str(e).splitlines()[0] if str(e) else traceback.format_exc().splitlines()[-2])
                    bad += 1
                except Exception as e:  # (a test that needs something the emulator's harness does not have)
                    verdict = "skipped (%s: %s)" % (type(e).__name__, str(e)[:80])
                ran += verdict == "ok"
                print("%-16s %-70s %s  (%.0f s)" % (title, name + (str(p) if p else ""), verdict, time.time() - t0), flush=True)
    print("serial kernels under the emulator: %d tests passed, %d mismatches" % (ran, bad))
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
