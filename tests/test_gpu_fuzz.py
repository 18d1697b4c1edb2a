"""Short runs of the differential fuzzers (tools/fuzz_decoders.py, tools/fuzz_encoders.py): every decoder variant against the oracle on
random mutations, truncations and capacity changes (status, error offset, plaintext); every encoder variant byte for byte on inputs of
many shapes."""
import pytest

pytestmark = pytest.mark.gpu


def test_decoders_agree_with_the_oracle_on_mutated_streams():
    from tools import fuzz_decoders
    assert fuzz_decoders.run(3000, 11) == 0


def test_zstd_and_container_decoders_agree_with_the_oracle_on_mutated_streams():
    from tools import fuzz_decoders
    assert fuzz_decoders.run(1500, 12, ("zstd", "lz4frame", "snappyframed")) == 0


def test_zstd_sequence_streams_damaged_at_their_end_decode_as_the_reference_decodes_them():
    """tools/fuzz_zstd_tail.py, a short run (round 6): frames damaged where their sequence bit streams are read LAST.  A sequence whose extra bits run past the stream's
    start is executed by the Java reader from what its wrapped shifts return -- the pipeline's sequence stage now reads the same bits (it used to hand such items to the
    one-kernel decoder, and the incremental reader, which has none behind it, refused a stream the reference reads: found by tools/fuzz_zstd_stream.py)."""
    from tools import fuzz_zstd_tail
    assert fuzz_zstd_tail.run(768, 14) == 0


def test_encoders_are_byte_identical_with_the_oracle_on_inputs_of_many_shapes():
    from tools import fuzz_encoders
    assert fuzz_encoders.run(400, 13) == 0


def test_incremental_zstd_stream_twins_agree_with_the_reference_reader_and_the_oracle_writer():
    """tools/fuzz_zstd_stream.py, a short run: streams valid, mutated, cut and extended, read through ZstdHipInputStream in random piece sizes against the
    reference's own ZstdInputStream transliterated (oracle/_ref/libref.so: skipped where that is absent), and random plaintexts written through
    ZstdHipOutputStream in random piece sizes against the oracle writer's bytes"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "libref.so")):
        pytest.skip("oracle/_ref/libref.so absent")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_zstd_stream.py"), "80", "25", "31"], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0 and "TOTAL MISMATCHES 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
