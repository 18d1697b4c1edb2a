#!/bin/bash
# Round 6 GPU calls, one parametrised script: tools/r06/call.sh <step> [<step> ...]
# Every step runs under its own timeout and writes under gpurun_out/r06/<step>*.
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
ks() {  # ks <tag> <bench args...>: per-kernel average times of one bench.py run -> $O/kstats_<tag>.txt
  local tag=$1; shift
  timeout 400 bash tools/kstats.sh r06_$tag "$@"
  mv gpurun_out/kstats_r06_$tag.txt $O/kstats_$tag.txt 2>/dev/null
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
    zstdtests)     # the Zstd decoder's GPU parity tests (pipeline, multi-block stages, streams)
      timeout 1200 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zstd_stream.py -m gpu -x -q 2>&1 | tail -5 ;;
    zstdprof)      # per-dispatch times of the Zstd section (K1 .. K5, the encoder's two kernels) -> $O/zstd_dispatches_<TAG>.txt
      timeout 900 bash tools/profile_zstd.sh r06_${TAG:-a} --no-cpu-baseline 2>&1 | tail -2 | cut -c1-1500
      cp gpurun_out/prof_r06_${TAG:-a}/keep/dispatches.txt $O/zstd_dispatches_${TAG:-a}.txt
      python - $O/zstd_dispatches_${TAG:-a}.txt <<'PY'
import sys, collections
t = collections.OrderedDict()
for l in open(sys.argv[1]).read().splitlines()[1:]:
    f = l.split()
    name = " ".join(f[:-4]); us = float(f[-4])
    t.setdefault(name, []).append(us)
for n, v in t.items():
    print("%-52s calls %4d  avg %9.1f us  min %9.1f  max %9.1f" % (n, len(v), sum(v) / len(v), min(v), max(v)))
PY
      ;;
    zstdfuzz)      # differential fuzz of the Zstd decoder (mutated frames: status, offset, plaintext against the oracle)
      timeout 900 python tools/fuzz_decoders.py ${N:-4000} 66 zstd 2>&1 | grep -v "^\[" | tail -8 | tee $O/fuzz_zstd.txt ;;
    seqwaves)      # the Zstd sequence stage at 1 / 2 / 4 wavefronts per workgroup: parity test, then the section's per-dispatch times for each
      timeout 600 python -m pytest tests/test_gpu_zstd.py -m gpu -x -q -k "wavefronts_per_workgroup" 2>&1 | tail -3
      for w in ${WAVES:-1 2 4}; do
        timeout 600 bash tools/profile_zstd.sh r06_w$w --no-cpu-baseline --zstd-seq-waves $w 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('waves $w:', {k:(v['decompress_GiBps'], v['java_frames_decompress_GiBps']) for k,v in r.items()})"
        grep sequences_lane gpurun_out/prof_r06_w$w/keep/dispatches.txt | awk '{print $2}' | tr '\n' ' '; echo
        cp gpurun_out/prof_r06_w$w/keep/dispatches.txt $O/zstd_dispatches_w$w.txt
      done 2>&1 | tee $O/seqwaves.txt ;;
    snappyfan)     # Snappy buffers beyond 64 KiB: the sub-blocks side by side -- parity (corpus manifest, encoder tests, mixed batch), then a 4 MB file per call
      timeout 1200 python -m pytest tests/test_gpu_lz4_snappy.py tests/test_gpu_corpus.py tests/test_gpu_hadoop.py tests/test_gpu_snappy_framed.py -m gpu -x -q 2>&1 | tail -4
      timeout 300 python tools/r06/big_snappy.py 2>&1 | grep -v amdgpu.ids | tee $O/big_snappy.txt ;;
    profweak)      # kernel-trace + HBM + issue counters of everything that is NOT the headline, on corpus data
      B="python bench.py --no-cpu-baseline --no-extra --no-legs --no-host-facing --no-sweep --steps 3 --warmup 1"
      for spec in ${SPECS:-lz4d snappyd lz4c snappyc zstdd zstdc}; do case $spec in
        lz4d)    timeout 900 bash tools/r06/counters.sh lz4_decompress_corpus $B --workload lz4_decompress --data corpus ;;
        snappyd) timeout 900 bash tools/r06/counters.sh snappy_decompress_corpus $B --workload snappy_decompress --data corpus ;;
        lz4c)    timeout 900 bash tools/r06/counters.sh lz4_compress_corpus $B --workload lz4_compress --data corpus --blocks 65536 ;;
        snappyc) timeout 900 bash tools/r06/counters.sh snappy_compress_corpus $B --workload snappy_compress --data corpus --blocks 65536 ;;
        zstdd)   timeout 900 bash tools/r06/counters.sh zstd_decompress_corpus python tools/zstd_batch_sizes.py 65536 ;;
        zstdc)   timeout 900 bash tools/r06/counters.sh zstd_compress_corpus python tools/r06/zstd_compress_run.py 32768 ;;
      esac; done 2>&1 | grep -v amdgpu.ids ;;
    hostblit)      # the host-pointer pipeline with its uploads / downloads by a copy kernel instead of hipMemcpyAsync (host.blit bits 0 / 1), three runs each
      for spec in ${SPECS:-"host.blit=0" "host.blit=2 host.blit_groups=128" "host.blit=2 host.blit_groups=32" "host.blit=1 host.blit_groups=128" "host.blit=3 host.blit_groups=128"}; do
        for rep in 1 2 3; do echo -n "$spec: "; timeout 300 python tools/host_path_rate.py 0 96 $spec 2>&1 | grep -v amdgpu.ids | tail -1; done
      done | tee $O/hostblit.txt ;;
    enctests)      # the LZ4 / Snappy encoders: parity tests (oracle bytes, manifest hashes, containers) + GPU fuzz
      timeout 1200 python -m pytest tests/test_gpu_lz4_snappy.py tests/test_gpu_corpus.py tests/test_gpu_hadoop.py tests/test_gpu_snappy_framed.py tests/test_gpu_lz4_frame.py -m gpu -x -q 2>&1 | tail -4
      timeout 900 python tools/fuzz_encoders.py ${N:-3000} 91 ${CODECS:-lz4 snappy} 2>&1 | tail -8 | tee $O/fuzz_encoders.txt ;;
    encrate)       # LZ4 / Snappy compress on the corpus batch: GiB/s as bench.py reports it
      for wl in ${WLS:-lz4_compress snappy_compress}; do for data in corpus fragments; do
        timeout 600 python bench.py --no-cpu-baseline --no-extra --no-legs --no-host-facing --no-sweep --steps 3 --warmup 1 --workload $wl --data $data --blocks 65536 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl $data', r['value'], 'GiB/s', r['roofline'].get('kernel_ms_avg'))"
      done; done | tee $O/encrate_${TAG:-a}.txt ;;
    lititems)      # the Zstd literal stage at 16 / 10 / 8 items per wavefront (zstd.decompress.lit_items): parity on the GPU, then the section's per-dispatch times
      for v in ${ITEMS:-16 10 8}; do
        timeout 600 bash tools/profile_zstd.sh r06_li$v --no-cpu-baseline --option zstd.decompress.lit_items=$v 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('lit_items $v:', {k:(v['decompress_GiBps'], v['java_frames_decompress_GiBps']) for k,v in r.items()})"
        grep literals_kernel gpurun_out/prof_r06_li$v/keep/dispatches.txt | awk '{print $(NF-3)}' | tr '\n' ' '; echo
      done 2>&1 | tee $O/lititems.txt ;;
    hadoopfuzz)    # differential fuzz of the Hadoop block-stream readers and the Snappy framed reader (mutated streams: status, offset, plaintext against the oracle)
      timeout 1200 python tools/fuzz_decoders.py ${N:-8000} 631 lz4hadoop,snappyhadoop,snappyframed 2>&1 | grep -v amdgpu.ids | tail -14 | tee $O/fuzz_hadoop.txt ;;
    memwaves)      # the Snappy encoder's memory tier at 3 / 2 / 1 / 0 wavefronts per workgroup (snappy.compress.mem_waves), corpus and fragments
      for data in corpus fragments; do for w in ${WAVES:-3 2 1 0}; do
        timeout 600 python bench.py --no-cpu-baseline --no-extra --no-legs --no-host-facing --no-sweep --steps 3 --warmup 1 --workload snappy_compress --data $data --blocks 65536 --option snappy.compress.mem_waves=$w 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('snappy_compress $data mem_waves $w:', r['value'], 'GiB/s  kernel ms', r['roofline'].get('kernel_ms_avg'))"
      done; done | tee $O/memwaves.txt ;;
    final)         # the pass that ships: suite, smoke, traffic.json, bench.py as the driver runs it, rocprofv3 summaries of both headline kernels, Zstd per-dispatch times, --gpus 2 on one device
      F=$O/final; rm -rf $F; mkdir -p $F
      timeout 1800 python -m pytest tests -m gpu -x -q > $F/pytest.log 2>&1; tail -2 $F/pytest.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
      timeout 1500 python tools/make_traffic_json.py $F/traffic.json > $F/traffic.log 2>&1; tail -1 $F/traffic.log | cut -c1-600; cp $F/traffic.json profiles/traffic.json  # (first: the bench line reports it only for the sources it was taken on)
      S=$(date +%s); timeout 1500 python bench.py > $F/bench_final.json 2> $F/bench_final.err; echo "bench.py wall: $(( $(date +%s) - S )) s" | tee -a $F/bench_final.err
      python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06/final/bench_final.json") if l.startswith("{")][-1])
print("value", r["value"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], "cpu", r["cpu_baseline"]["value"])
for k in sorted(r):
    if k.startswith("value_") or k in ("mixed_ok", "end_to_end", "single_block_us", "mixed_8x_batch"):
        print(k, r[k] if not isinstance(r[k], dict) else {a: b for a, b in r[k].items() if a != "what"})
for k in ("lz4_corpus", "snappy_corpus", "zstd_corpus"):
    print(k, {a: b for a, b in r["extra"][k].get("roofline", {}).items() if a not in ("traffic_source", "traffic_per_kernel")})
PY
      timeout 700 bash tools/profile.sh r06final_lz4 --steps 5 --warmup 2 --no-legs --no-host-facing > $F/profile_lz4_summary.txt 2>&1
      cp gpurun_out/prof_r06final_lz4/keep/*kernel_stats.csv $F/lz4_kernel_stats.csv 2>/dev/null
      timeout 700 bash tools/profile.sh r06final_snappy --steps 5 --warmup 2 --no-legs --no-host-facing --workload snappy_decompress > $F/profile_snappy_summary.txt 2>&1
      cp gpurun_out/prof_r06final_snappy/keep/*kernel_stats.csv $F/snappy_kernel_stats.csv 2>/dev/null
      timeout 500 bash tools/profile_zstd.sh r06finalz --no-cpu-baseline > $F/zstd_line.txt 2>&1
      cp gpurun_out/prof_r06finalz/keep/dispatches.txt $F/zstd_dispatches.txt; cp gpurun_out/prof_r06finalz/keep/*kernel_stats.csv $F/zstd_kernel_stats.csv
      ACHIP_BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 2 --blocks 65536 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $F/n2.json 2> $F/n2.err; grep -c '^{' $F/n2.json
      ;;
    spread)        # run-to-run spread of the headline line on one box
      for i in 1 2 3 4 5 6 7 8 9; do timeout 300 python bench.py --no-cpu-baseline --no-extra --no-legs --no-host-facing 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('run $i', r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_avg'])"; done | tee $O/headline_spread.txt ;;
    sizesweep)     # the default decoders against the batch size, final library
      for w in lz4_decompress snappy_decompress; do for data in fragments corpus; do for n in 1024 4096 16384 65536 262144; do
        timeout 300 python bench.py --workload $w --data $data --blocks $n --pool 512 --no-extra --no-legs --no-host-facing --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w $data %7d x 64 KiB: %8.1f GiB/s  %8.3f ms' % ($n, r['value'], r['ms_per_step']))"
      done; done; done 2>&1 | tee $O/sizesweep.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
