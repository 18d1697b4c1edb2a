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


def test_encoders_are_byte_identical_with_the_oracle_on_inputs_of_many_shapes():
    from tools import fuzz_encoders
    assert fuzz_encoders.run(400, 13) == 0
