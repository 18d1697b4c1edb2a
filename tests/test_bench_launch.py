"""bench.py's entry point: `--gpus N` must MEAN N.

Runs here without a GPU: ACHIP_BENCH_STUB_DEVICE=1 keeps the whole control plane of an N-rank run -- bench.py starting its own ranks under
torch.distributed.run when no launcher did, the gloo rendezvous on 127.0.0.1, shard_for_rank, barrier, MAX over ranks, the per-rank gather,
ONE line from rank 0 -- and replaces the kernel with a host memcpy (the line says "stub").  On a GPU box the same entry point runs the real
thing (tests/test_gpu_corpus.py::test_bench_gpus_2_on_one_device, ACHIP_BENCH_SHARE_DEVICE=1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env, *argv, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)


def json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_gpus_2_starts_two_ranks_by_itself_and_prints_one_line():
    r = run_bench({"ACHIP_BENCH_STUB_DEVICE": "1"}, "--gpus", "2", "--blocks", "4096", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["stub"] is True and "stub" in line["data"]
    assert len(line["config"]["per_rank_seconds"]) == 2            # both ranks reported
    assert line["config"]["blocks_per_gpu"] == 4096                # weak scaling: --blocks per GPU
    # the line carries the WHOLE metric at N > 1: Snappy, Zstd configs[3] and the mixed corpus batch configs[4] as
    # aggregates over the ranks, each with its per-rank entries
    for key in ("value_snappy", "value_zstd", "value_zstd_corpus", "value_mixed"):
        assert key in line, key
    for leg in ("snappy", "zstd", "zstd_corpus", "zstd_java_frames", "mixed"):
        assert {p["rank"] for p in line["legs"][leg]["per_rank"]} == {0, 1}, leg
    mixed = line["legs"]["mixed"]
    assert line["mixed_ok"] is True and mixed["manifest_lines"] == 668
    slices = sorted(p["slice"] for p in mixed["per_rank"])           # achip_partition_blocks: contiguous, complete, balanced by bytes
    assert slices[0][0] == 0 and slices[0][1] == slices[1][0] and slices[1][1] == mixed["items"]
    by = [p["bytes"] for p in mixed["per_rank"]]
    assert abs(by[0] - by[1]) < 0.1 * sum(by)


def test_gpus_1_stays_one_process():
    r = run_bench({"ACHIP_BENCH_STUB_DEVICE": "1"}, "--blocks", "1024", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1


def test_a_world_of_another_size_is_refused():
    """a launcher that started 3 ranks for `--gpus 2` (or one rank for `--gpus 8`) must not produce a line under the requested N's name"""
    r = run_bench({"ACHIP_BENCH_STUB_DEVICE": "1", "WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert r.returncode != 0 and json_lines(r.stdout) == []
    assert "WORLD_SIZE=3" in r.stderr
    r = run_bench({"ACHIP_BENCH_STUB_DEVICE": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "8")
    assert r.returncode != 0 and json_lines(r.stdout) == []


import pytest


@pytest.mark.gpu
def test_bench_gpus_2_on_one_device():
    """The real thing on a one-GPU box: `python bench.py --gpus 2` with no launcher on the command line, both ranks on cuda:0
    (ACHIP_BENCH_SHARE_DEVICE=1: a path check, labelled as such in the line) -- one line, n_gpus 2, both ranks verified bit-exact."""
    r = run_bench({"ACHIP_BENCH_SHARE_DEVICE": "1"}, "--gpus", "2", "--blocks", "16384", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--zstd-frames", "4096",
                  "--mixed-copies", "1", timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and len(line["per_rank"]) == 2 and {p["rank"] for p in line["per_rank"]} == {0, 1}
    assert "ALL RANKS ON ONE DEVICE" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    # ... and the rest of the metric from both ranks: every leg verified byte-exact on every rank before it was timed
    assert line["mixed_ok"] is True and all(p["mismatches"] == 0 and p["items"] > 0 for p in line["legs"]["mixed"]["per_rank"])
    for key in ("value_snappy", "value_zstd", "value_zstd_corpus", "value_zstd_java_frames", "value_mixed"):
        assert line[key] > 0, key
    assert {p["rank"] for p in line["legs"]["zstd_corpus"]["per_rank"]} == {0, 1}


@pytest.mark.gpu
def test_bench_gpus_2_without_two_devices_is_an_error():
    """... and without the path-check switch two ranks on a one-GPU box must fail, not report"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("the box has two devices")
    r = run_bench({}, "--gpus", "2", "--blocks", "4096", "--steps", "1", "--warmup", "0", "--no-extra", "--no-cpu-baseline", timeout=900)
    assert r.returncode != 0 and json_lines(r.stdout) == []
