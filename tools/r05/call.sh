#!/bin/bash
# Round 5 GPU calls, one parametrised script: tools/r05/call.sh <step> [<step> ...]
# Every step runs under its own timeout and writes under gpurun_out/r05/<step>*.
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
ks() {  # ks <tag> <bench args...>: per-kernel average times of one bench.py run -> $O/kstats_<tag>.txt
  local tag=$1; shift
  timeout 400 bash tools/kstats.sh r05_$tag "$@"
  mv gpurun_out/kstats_r05_$tag.txt $O/kstats_$tag.txt 2>/dev/null
  echo "--- $tag"; cat $O/kstats_$tag.txt
}
line() {  # line <label>: one bench line on stdin -> the keys worth reading
  python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'value', r.get('value'), 'frac', r.get('roofline',{}).get('frac'), {k:v for k,v in r.items() if k.startswith('value_') or k in ('mixed_ok','end_to_end','single_block_us')})
"
}
for step in "$@"; do
  echo "===== $step at $(date +%T)"
  case $step in
    newtests)      # this round's new GPU tests: several contexts in one process, the host-pointer pipeline, bench --gpus 2 with all legs
      timeout 900 python -m pytest tests/test_gpu_corpus.py tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -5 ;;
    hostfacing)    # end_to_end + single_block_us (the headline's line without extras)
      timeout 600 python bench.py --no-extra --no-cpu-baseline --no-legs --steps 5 --warmup 2 2> $O/hostfacing.err | tee $O/hostfacing.json | line hostfacing; tail -3 $O/hostfacing.err ;;
    n2)            # --gpus 2 on one device with every leg
      ACHIP_BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 2 --blocks 65536 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/n2.json 2> $O/n2.err; tail -3 $O/n2.err; cat $O/n2.json | line n2 ;;
    hostsweep)     # the pageable host path over copy threads x chunk size, with its stage times
      timeout 600 python tools/host_path_rate.py 0,4,8,16,32 96,192,384 2>&1 | grep -v amdgpu.ids | tee $O/hostsweep.txt ;;
    encfuzz)       # the Zstd encoder after the entropy helpers' rewrite: GPU tests of the encoder + differential fuzz (bytes against the oracle)
      timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zstd_stream.py tests/test_gpu_corpus.py -m gpu -x -q 2>&1 | tail -3
      timeout 900 python tools/fuzz_encoders.py 2>&1 | tail -12 | tee $O/fuzz_encoders.txt ;;
    zstream)       # the incremental Zstd reader + the encoder after the FSE table rewrite
      timeout 1200 python -m pytest tests/test_gpu_zstd_stream.py tests/test_gpu_zstd.py -m gpu -x -q 2>&1 | tail -15
      timeout 600 python tools/fuzz_encoders.py 1500 77 zstd 2>&1 | tail -5 ;;
    bench)         # the default bench.py as the driver runs it, with its wall clock
      S=$(date +%s); timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py wall: $(( $(date +%s) - S )) s rc=$?" | tee -a $O/bench.err; tail -3 $O/bench.err
      python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r05/bench.json") if l.startswith("{")][-1])
print("value", r["value"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], "cpu", r.get("cpu_baseline", {}).get("value"))
for k in sorted(r):
    if k.startswith("value_") or k in ("mixed_ok", "end_to_end", "single_block_us"):
        print(k, r[k] if not isinstance(r[k], dict) else {a: b for a, b in r[k].items() if a != "what"})
PY
      ;;
    tworounds)     # the two-pass decoders on a batch of two rounds (524288 blocks) beside the one-round batch (verdict 4c)
      for wl in lz4_decompress snappy_decompress; do for nb in 262144 524288; do
        timeout 600 python bench.py --no-cpu-baseline --no-extra --no-legs --no-host-facing --steps 5 --warmup 2 --workload $wl --data corpus --blocks $nb 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl corpus blocks $nb', r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_avg'])"
      done; done | tee $O/tworounds.txt ;;
    latency)       # the ring decoders' latency class: parity tests + one block per call
      timeout 900 python -m pytest tests/test_gpu_lz4_snappy.py tests/test_gpu_corpus.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
      timeout 300 python tools/single_block_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/single_block_latency.txt ;;
    snappywave)    # the wavefront-per-block Snappy parser: parity tests of everything that reaches it, fuzz, one block per call, the mixed batch by bucket
      timeout 1200 python -m pytest tests/test_gpu_lz4_snappy.py tests/test_gpu_snappy_framed.py tests/test_gpu_hadoop.py tests/test_gpu_corpus.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
      ( timeout 500 python tools/fuzz_decoders.py 8000 61 snappy
        timeout 400 python tools/fuzz_decoders.py 2000 62 snappy big ) 2>&1 | grep -v "^\[" | tee $O/fuzz_snappy_wave.txt
      timeout 300 python tools/single_block_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/single_block_latency.txt
      timeout 300 python tools/mixed_breakdown.py 2>&1 | grep -v amdgpu.ids | tee $O/mixed_breakdown.txt ;;
    mixlanes)      # a mixed batch's buckets on helper contexts: the corpus tests, the bench leg with and without
      timeout 900 python -m pytest tests/test_gpu_corpus.py tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -4
      for c in 1 0; do ACHIP_MIXED_CONCURRENT=$c timeout 600 python bench.py --no-extra --no-cpu-baseline --no-host-facing --blocks 65536 --steps 3 --warmup 1 2> $O/mixlanes_$c.err | tee $O/mixlanes_$c.json | line "mixed.concurrent=$c"; done ;;
    hwqueues)      # do the helper streams of a mixed batch share hardware queues?  (ROCm maps a process's streams onto GPU_MAX_HW_QUEUES queues, 4 by default)
      for q in ${HWQ:-4 8}; do GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --no-extra --no-cpu-baseline --no-host-facing --blocks 65536 --steps 3 --warmup 1 2> $O/hwq_$q.err | tee $O/hwq_$q.json | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('GPU_MAX_HW_QUEUES=$q', r['legs']['mixed']['per_rank'][0]['seconds'], 's', r['value_mixed'], 'GiB/s')"; done ;;
    groupsweep)    # ring decoders: lanes per block against batch size, blocks of 256 KiB (what the frame and Hadoop readers hand over)
      for data in fragments corpus; do for n in 4096 16384 65536; do for g in 4 16 64; do
        timeout 300 python bench.py --workload lz4_decompress --data $data --blocks $n --block-size 262144 --pool 128 --group $g --variant 1 --no-extra --no-legs --no-host-facing --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$data blocks $n x 256 KiB, $g lanes per block:', r['value'], 'GiB/s')"
      done; done; done 2>&1 | tee $O/groupsweep.txt ;;
    groupsweep2)   # ... and blocks of 64 KiB, the group sizes between
      for spec in "8192 65536" "16384 65536" "32768 65536" "65536 65536" "131072 65536" "8192 262144" "32768 262144" "1024 4194304"; do set -- $spec; for g in 4 8 16 32 64; do
        timeout 300 python bench.py --workload lz4_decompress --data fragments --blocks $1 --block-size $2 --pool 64 --group $g --variant 1 --no-extra --no-legs --no-host-facing --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fragments blocks $1 x $2, $g lanes per block:', r['value'], 'GiB/s')"
      done; done 2>&1 | tee $O/groupsweep2.txt ;;
    ringgroups)    # lanes per block by the batch size + the LZ4 frame reader's listed blocks through the rings: parity, the container section, the block API at mid sizes
      timeout 1200 python -m pytest tests/test_gpu_lz4_frame.py tests/test_gpu_hadoop.py tests/test_gpu_lz4_snappy.py tests/test_gpu_corpus.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
      timeout 600 python bench.py --section lz4frame --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for k,v in sorted(r.items()): print(k, {a:b for a,b in v.items() if 'GiBps' in a or a in ('decoder',)} if isinstance(v,dict) else v)" | tee $O/lz4frame_section.txt
      for spec in "1024 4194304" "4096 262144" "16384 65536" "65536 65536"; do set -- $spec
        timeout 300 python bench.py --workload lz4_decompress --data fragments --blocks $1 --block-size $2 --pool 64 --no-extra --no-legs --no-host-facing --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default decoder, fragments blocks $1 x $2:', r['value'], 'GiB/s')"
      done | tee $O/ringgroups_block_api.txt
      timeout 600 python tools/fuzz_decoders.py 4000 71 lz4frame 2>&1 | grep -v "^\[" | tail -3 ;;
    fewkernels)    # the wavefront-per-block parsers and the executor on few text blocks, per kernel
      for spec in "1 65536" "1024 65536" "64 4194304"; do set -- $spec
        rm -rf $O/fk; rocprofv3 --kernel-trace --stats --output-format csv -d $O/fk -o s -- python tools/few_blocks_kernels.py $1 $2 2>/dev/null | grep "two passes"
        python - $(find $O/fk -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "achip" in row["Name"] and ("parse" in row["Name"] or "execute" in row["Name"]):
        print("   %-60s calls=%s avg_us=%.1f" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
      done 2>&1 | tee $O/fewkernels.txt; rm -rf $O/fk ;;
    parsesweep)    # the two parsers of the two-pass decoders against the batch size (corpus blocks of 64 KiB)
      for w in lz4_decompress snappy_decompress; do for n in 4096 8192 16384 32768 65536 131072; do for ps in 1 2; do
        timeout 300 python bench.py --workload $w --data corpus --blocks $n --variant 7 --parse $ps --no-extra --no-legs --no-host-facing --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w corpus blocks $n x 64 KiB, parse $ps:', r['value'], 'GiB/s')"
      done; done; done 2>&1 | tee $O/parsesweep.txt ;;
    hybridsweep)   # (measured, removed: needs the parse mode 3 of that experiment) both parsers side by side: the wavefront parser's share against the batch size (corpus blocks of 64 KiB)
      for w in lz4_decompress snappy_decompress; do for n in 24576 32768 65536 131072 262144; do for share in 8192 16384 24576 32768 49152; do
        [ $share -ge $n ] && continue
        timeout 300 python bench.py --workload $w --data corpus --blocks $n --variant 7 --parse 3 --hybrid-wave-blocks $share --no-extra --no-legs --no-host-facing --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w corpus blocks $n x 64 KiB, wavefront parser takes $share:', r['value'], 'GiB/s')"
      done; done; done 2>&1 | tee $O/hybridsweep.txt ;;
    final)         # the pass that ships: suite, smoke, bench.py as the driver runs it, rocprofv3 summaries of BOTH headline kernels, traffic.json, Zstd per-dispatch times, --gpus 2 on one device
      F=$O/final; rm -rf $F; mkdir -p $F
      timeout 1800 python -m pytest tests -m gpu -x -q > $F/pytest.log 2>&1; tail -2 $F/pytest.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
      timeout 600 python tools/make_traffic_json.py $F/traffic.json > $F/traffic.log 2>&1; tail -1 $F/traffic.log | cut -c1-400; cp $F/traffic.json profiles/traffic.json  # (first: the bench line reports it only for the sources it was taken on)
      S=$(date +%s); timeout 1500 python bench.py > $F/bench_final.json 2> $F/bench_final.err; echo "bench.py wall: $(( $(date +%s) - S )) s" | tee -a $F/bench_final.err
      python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r05/final/bench_final.json") if l.startswith("{")][-1])
print("value", r["value"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], "cpu", r["cpu_baseline"]["value"])
for k in sorted(r):
    if k.startswith("value_") or k in ("mixed_ok", "end_to_end", "single_block_us"):
        print(k, r[k] if not isinstance(r[k], dict) else {a: b for a, b in r[k].items() if a != "what"})
PY
      timeout 700 bash tools/profile.sh r05final_lz4 --steps 5 --warmup 2 --no-legs --no-host-facing > $F/profile_lz4_summary.txt 2>&1
      cp gpurun_out/prof_r05final_lz4/keep/*kernel_stats.csv $F/lz4_kernel_stats.csv 2>/dev/null
      timeout 700 bash tools/profile.sh r05final_snappy --steps 5 --warmup 2 --no-legs --no-host-facing --workload snappy_decompress > $F/profile_snappy_summary.txt 2>&1
      cp gpurun_out/prof_r05final_snappy/keep/*kernel_stats.csv $F/snappy_kernel_stats.csv 2>/dev/null
      timeout 500 bash tools/profile_zstd.sh r05finalz --no-cpu-baseline > $F/zstd_line.txt 2>&1
      cp gpurun_out/prof_r05finalz/keep/dispatches.txt $F/zstd_dispatches.txt; cp gpurun_out/prof_r05finalz/keep/*kernel_stats.csv $F/zstd_kernel_stats.csv
      ACHIP_BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 2 --blocks 65536 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $F/n2.json 2> $F/n2.err; grep -c '^{' $F/n2.json
      ;;
    groupsweep3)   # the same for Snappy
      for spec in "1024 4194304" "4096 262144" "8192 65536" "16384 65536" "32768 65536"; do set -- $spec; for g in 4 16 64; do
        timeout 300 python bench.py --workload snappy_decompress --data fragments --blocks $1 --block-size $2 --pool 64 --group $g --variant 1 --no-extra --no-legs --no-host-facing --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('snappy fragments blocks $1 x $2, $g lanes per block:', r['value'], 'GiB/s')"
      done; done 2>&1 | tee $O/groupsweep3.txt ;;
    spread)        # run-to-run spread of the headline line on one box
      for i in 1 2 3 4 5 6 7 8 9; do timeout 300 python bench.py --no-cpu-baseline --no-extra --no-legs --no-host-facing 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('run $i', r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_avg'])"; done | tee $O/headline_spread.txt ;;
    sizesweep)     # the default decoders against the batch size, final library: LZ4 / Snappy blocks of 64 KiB (fragments, corpus), Zstd frames of 128 KiB
      for w in lz4_decompress snappy_decompress; do for data in fragments corpus; do for n in 1024 4096 16384 65536 262144; do
        timeout 300 python bench.py --workload $w --data $data --blocks $n --pool 512 --no-extra --no-legs --no-host-facing --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w $data %7d x 64 KiB: %8.1f GiB/s  %8.3f ms' % ($n, r['value'], r['ms_per_step']))"
      done; done; done 2>&1 | tee $O/sizesweep.txt
      timeout 300 python tools/zstd_batch_sizes.py 1024 4096 16384 65536 2>&1 | grep -v amdgpu | tee -a $O/sizesweep.txt
      timeout 300 python tools/zstd_batch_sizes.py --fragments 1024 4096 16384 65536 2>&1 | grep -v amdgpu | tee -a $O/sizesweep.txt ;;
    fuzz)          # differential fuzz of the decoders (status, offset, plaintext) against the oracle on the GPU, after this round's routing changes
      ( timeout 700 python tools/fuzz_decoders.py 20000 51 lz4,snappy
        timeout 500 python tools/fuzz_decoders.py 3000 52 lz4,snappy big
        timeout 900 python tools/fuzz_decoders.py 6000 53 zstd
        timeout 600 python tools/fuzz_decoders.py 3000 54 zstd big
        timeout 600 python tools/fuzz_decoders.py 6000 55 lz4frame,snappyframed ) 2>&1 | grep -v "^\[" | tee $O/fuzz_decoders.txt ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log ;;
    zstd)          # the Zstd section + per-kernel times
      timeout 500 bash tools/profile_zstd.sh r05zstd_$RANDOM --no-cpu-baseline > $O/zstd_line.txt 2>&1
      for d in gpurun_out/prof_r05zstd_*; do cp $d/keep/dispatches.txt $O/zstd_dispatches.txt; cp $d/keep/*kernel_stats.csv $O/zstd_kernel_stats.csv; done
      tail -c 1500 $O/zstd_line.txt
      cat $O/zstd_dispatches.txt | awk 'NR>1{t[$1]+=$2; n[$1]++} END{for(k in t) printf "%-48s n=%d total_us=%.0f\n", k, n[k], t[k]}' | sort ;;
    *) echo "unknown step $step" ;;
  esac
done
