#!/bin/bash
# Round 3, the pass that ships (run it again after every change of a default): the full GPU suite, bench.py as the driver runs it, the
# rocprofv3 --kernel-trace --stats summary + PMC passes of the same command (tools/profile.sh), roofline.traffic with the kernel sources' hash
# (tools/make_traffic_json.py), per-dispatch times of the Zstd pipeline, and the N = 2 bench path on one device.  -> gpurun_out/r03final/
export TMPDIR=/tmp
O=gpurun_out/r03final
rm -rf $O; mkdir -p $O
T0=$(date +%s)
stamp() { echo "== $1 at +$(( $(date +%s) - T0 )) s" | tee -a $O/timeline.txt; }

stamp "gpu tests"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log | tee -a $O/timeline.txt

stamp "smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/timeline.txt

stamp "bench.py (defaults)"
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
python - <<'PY' | tee -a gpurun_out/r03final/timeline.txt
import json
r = json.loads([l for l in open("gpurun_out/r03final/bench_final.json") if l.startswith("{")][-1])
print("value", r["value"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"].get("value_1GiB_sample"))
for k in ("value_corpus", "value_snappy", "value_snappy_corpus", "value_zstd", "value_zstd_corpus"):
    print(k, r.get(k))
PY

stamp "profile (kernel-trace stats + pmc passes) of the headline command"
timeout 600 bash tools/profile.sh r03final --steps 5 --warmup 2 > $O/profile_summary.txt 2>&1
cp gpurun_out/prof_r03final/keep/*kernel_stats.csv $O/ 2>/dev/null

stamp "traffic.json"
timeout 400 python tools/make_traffic_json.py $O/traffic.json > $O/traffic.log 2>&1
tail -1 $O/traffic.log | cut -c1-300 | tee -a $O/timeline.txt

stamp "zstd per-dispatch"
timeout 400 bash tools/profile_zstd.sh r03zstd --no-cpu-baseline > $O/zstd_line.txt 2>&1
cp gpurun_out/prof_r03zstd/keep/dispatches.txt $O/zstd_dispatches.txt 2>/dev/null
cp gpurun_out/prof_r03zstd/keep/*kernel_stats.csv $O/zstd_kernel_stats.csv 2>/dev/null

stamp "N = 2 path on one device"
ACHIP_BENCH_SHARE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --blocks 65536 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/n2.json 2> $O/n2.err
grep -c '^{' $O/n2.json | tee -a $O/timeline.txt
stamp "done"
