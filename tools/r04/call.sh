#!/bin/bash
# Round 4 GPU calls, one parametrised script (replaces round 3's tools/r03_call*.sh): tools/r04/call.sh <step> [<step> ...]
# Every step runs under its own timeout and writes under gpurun_out/r04/<step>*.
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
ks() {  # ks <tag> <bench args...>: per-kernel average times of one bench.py run -> $O/kstats_<tag>.txt
  local tag=$1; shift
  timeout 400 bash tools/kstats.sh r04_$tag "$@"
  mv gpurun_out/kstats_r04_$tag.txt $O/kstats_$tag.txt 2>/dev/null
  echo "--- $tag"; cat $O/kstats_$tag.txt
}
for step in "$@"; do
  echo "===== $step at $(date +%T)"
  case $step in
    base_corpus)   # per-kernel times of the two-pass decoders on the corpus batch
      ks lz4_corpus --workload lz4_decompress --data corpus --steps 5 --warmup 2
      ks snappy_corpus --workload snappy_decompress --data corpus --steps 5 --warmup 2 ;;
    base_zstd)
      timeout 500 bash tools/profile_zstd.sh r04zstd_$RANDOM --no-cpu-baseline > $O/zstd_line.txt 2>&1
      for d in gpurun_out/prof_r04zstd_*; do cp $d/keep/dispatches.txt $O/zstd_dispatches.txt; cp $d/keep/*kernel_stats.csv $O/zstd_kernel_stats.csv; done
      tail -c 1500 $O/zstd_line.txt
      cat $O/zstd_dispatches.txt | awk 'NR>1{t[$1]+=$2; n[$1]++} END{for(k in t) printf "%-48s n=%d total_us=%.0f\n", k, n[k], t[k]}' | sort ;;
    containers)    # the container extras with their CPU legs (bench.py --section lz4frame / zstdstream)
      timeout 900 python bench.py --section lz4frame > $O/containers.json 2> $O/containers.err; tail -c 300 $O/containers.err
      timeout 900 python bench.py --section zstdstream > $O/zstdstream.json 2> $O/zstdstream.err; tail -c 300 $O/zstdstream.err
      python - <<'PY'
import json
for f in ("gpurun_out/r04/containers.json", "gpurun_out/r04/zstdstream.json"):
    for l in open(f):
        if l.startswith("{"):
            for k, v in json.loads(l).items():
                print(k, {a: b for a, b in v.items() if "GiBps" in a or a in ("cpu_threads", "frames", "ratio")})
PY
      ;;
    pmc_zseq)      # SQ counters of the Zstd pipeline's kernels (bench.py --section zstd)
      bash tools/pmc_any.sh r04zs1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" python bench.py --section zstd --no-cpu-baseline
      bash tools/pmc_any.sh r04zs2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM" python bench.py --section zstd --no-cpu-baseline
      grep "sequences\|execute2\|literals\|pipe_parse" gpurun_out/pmc_r04zs1.txt gpurun_out/pmc_r04zs2.txt | sed 's/gpurun_out.pmc_r04//' | cut -c1-200
      cp gpurun_out/pmc_r04zs1.txt gpurun_out/pmc_r04zs2.txt $O/ ;;
    pmc_twopass)   # SQ counters of the two-pass decoders' kernels on the corpus batch
      for wl in lz4_decompress snappy_decompress; do
        bash tools/pmc.sh r04tp1_$wl "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" --workload $wl --data corpus --steps 3 --warmup 1
        bash tools/pmc.sh r04tp2_$wl "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" --workload $wl --data corpus --steps 3 --warmup 1
        grep "parse2\|execute2" gpurun_out/pmc_r04tp1_$wl.txt gpurun_out/pmc_r04tp2_$wl.txt | sed 's/gpurun_out.pmc_r04//; s/(achip::BatchArgs.*) *SQ/ SQ/' | cut -c1-150
        cat gpurun_out/pmc_r04tp1_$wl.txt gpurun_out/pmc_r04tp2_$wl.txt > $O/pmc_twopass_$wl.txt
      done ;;
    final)         # the pass that ships: suite, smoke, bench.py as the driver runs it, rocprofv3 summaries of BOTH headline kernels, traffic.json, Zstd per-dispatch times, --gpus 2 on one device
      F=$O/final; rm -rf $F; mkdir -p $F
      timeout 1500 python -m pytest tests -m gpu -x -q > $F/pytest.log 2>&1; tail -2 $F/pytest.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
      timeout 1500 python bench.py > $F/bench_final.json 2> $F/bench_final.err
      python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r04/final/bench_final.json") if l.startswith("{")][-1])
print("value", r["value"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], "cpu", r["cpu_baseline"]["value"])
for k in ("value_corpus", "value_snappy", "value_snappy_corpus", "value_zstd", "value_zstd_corpus", "value_zstd_stream_corpus"):
    print(k, r.get(k))
PY
      timeout 700 bash tools/profile.sh r04final_lz4 --steps 5 --warmup 2 > $F/profile_lz4_summary.txt 2>&1
      cp gpurun_out/prof_r04final_lz4/keep/*kernel_stats.csv $F/lz4_kernel_stats.csv 2>/dev/null
      timeout 700 bash tools/profile.sh r04final_snappy --steps 5 --warmup 2 --workload snappy_decompress > $F/profile_snappy_summary.txt 2>&1
      cp gpurun_out/prof_r04final_snappy/keep/*kernel_stats.csv $F/snappy_kernel_stats.csv 2>/dev/null
      timeout 600 python tools/make_traffic_json.py $F/traffic.json > $F/traffic.log 2>&1; tail -1 $F/traffic.log | cut -c1-400
      timeout 500 bash tools/profile_zstd.sh r04finalz --no-cpu-baseline > $F/zstd_line.txt 2>&1
      cp gpurun_out/prof_r04finalz/keep/dispatches.txt $F/zstd_dispatches.txt; cp gpurun_out/prof_r04finalz/keep/*kernel_stats.csv $F/zstd_kernel_stats.csv
      ACHIP_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --blocks 65536 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $F/n2.json 2> $F/n2.err; grep -c '^{' $F/n2.json
      ;;
    sweep)         # the Random(301) sweep + the Snappy / LZ4 headline (ring decoders)
      timeout 600 python bench.py --section sweep 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        for k, v in json.loads(l).items(): print(k, v['decompress_GiBps'], v.get('decompress_hbm_frac'))
" | tee $O/sweep.txt
      for wl in lz4_decompress snappy_decompress; do for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl', r['value'], r['roofline']['frac'])"; done; done | tee -a $O/sweep.txt ;;
    mixed)         # the mixed batch (corpus + fragments side by side): auto mode beside the forced decoders
      for wl in lz4_decompress snappy_decompress; do for v in "" "--variant 7" "--variant 1"; do
        timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 --workload $wl --data mixed $v 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl mixed [$v]', r['value'], r['config']['decoder'][:60])"
      done; done | tee $O/mixed.txt ;;
    choice)        # what auto mode picks, and what it makes, per kind of batch
      for wl in lz4_decompress snappy_decompress; do for d in "" "--data corpus" "--data fragments" "--data mixed" "--ratio 0.25" "--ratio 0.1"; do
        timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 --workload $wl $d 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl [$d]', r['value'], r['config']['decoder'][:70])"
      done; done | tee $O/choice.txt ;;
    fuzz)          # differential fuzz of the decoders (status, offset, plaintext) and the encoders (bytes) against the oracle, on the GPU
      ( timeout 700 python tools/fuzz_decoders.py 20000 41 lz4,snappy
        timeout 500 python tools/fuzz_decoders.py 3000 42 lz4,snappy big
        timeout 900 python tools/fuzz_decoders.py 6000 43 zstd
        timeout 600 python tools/fuzz_decoders.py 3000 44 zstd big
        timeout 600 python tools/fuzz_decoders.py 6000 45 lz4frame,snappyframed ) 2>&1 | grep -v "^\[" | tee $O/fuzz_decoders.txt
      timeout 900 python tools/fuzz_encoders.py 2>&1 | tail -12 | tee $O/fuzz_encoders.txt ;;
    kzs)           # per-kernel totals of the Zstd multi-block stream section
      export TMPDIR=/tmp; rm -rf gpurun_out/ks_zs; mkdir -p gpurun_out/ks_zs
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_zs -o s -- python bench.py --section zstdstream --no-cpu-baseline > gpurun_out/ks_zs/log.txt 2>&1
      python - $(find gpurun_out/ks_zs -name '*kernel_stats.csv' | head -1) <<'PY' | tee $O/kstats_zstdstream.txt
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "achip" in row["Name"] and float(row["TotalDurationNs"]) > 2e5:
        print("%-80s calls=%s avg_ms=%.3f total_ms=%.1f" % (row["Name"][:80], row["Calls"], float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"]) / 1e6))
PY
      grep '^{' gpurun_out/ks_zs/log.txt | cut -c1-900; rm -rf gpurun_out/ks_zs ;;
    traffic_zstd)  # HBM bytes per kernel of the Zstd section (FETCH_SIZE x 2: gfx950 counts 32-byte units as 64 -- profiles/r03_counter_calibration.txt; WRITE_SIZE as counted), in separate passes
      bash tools/pmc_any.sh r04zf "FETCH_SIZE" python bench.py --section zstd --no-cpu-baseline
      bash tools/pmc_any.sh r04zw "WRITE_SIZE" python bench.py --section zstd --no-cpu-baseline
      cat gpurun_out/pmc_r04zf.txt gpurun_out/pmc_r04zw.txt | grep "zstd" | tee $O/zstd_traffic.txt ;;
    benchtime)     # the wall clock of the default bench.py run
      S=$(date +%s); python bench.py > $O/bench_timed.json 2> $O/bench_timed.err; echo "bench.py wall: $(( $(date +%s) - S )) s" | tee -a $O/bench_timed.err ;;
    tests)
      timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log ;;
    *) echo "unknown step $step" ;;
  esac
done
