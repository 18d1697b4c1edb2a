"""A short run of the differential decoder fuzz (tools/fuzz_decoders.py): every LZ4 / Snappy decoder variant against the oracle on
random mutations, truncations and capacity changes -- status, error offset and plaintext."""
import pytest

pytestmark = pytest.mark.gpu


def test_decoders_agree_with_the_oracle_on_mutated_streams():
    from tools import fuzz_decoders
    assert fuzz_decoders.run(3000, 11) == 0


def test_zstd_and_container_decoders_agree_with_the_oracle_on_mutated_streams():
    from tools import fuzz_decoders
    assert fuzz_decoders.run(1500, 12, ("zstd", "lz4frame", "snappyframed")) == 0
