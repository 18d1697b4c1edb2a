#!/bin/bash
# Round 3, GPU call 1 (profiles/r03_notes.md has the budget table): ship-blockers first, then the measure-or-delete pass over the variants
# round 2 left unmeasured, the N > 1 path check, the counter calibration.  Every step under its own timeout; everything lands in gpurun_out/r03c1/.
export TMPDIR=/tmp
O=gpurun_out/r03c1
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== $1 at +$(( $(date +%s) - T0 )) s" | tee -a $O/timeline.txt; }
B="python bench.py --no-cpu-baseline --no-sweep"

stamp "gpu tests (experimental variants included)"
ACHIP_TEST_EXPERIMENTAL=1 timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log | tee -a $O/timeline.txt

stamp "lz4 corpus: auto / forced two-pass / parts / 8 KiB window"
for v in "" "--variant 7" "--variant 7 --exec-variant 302" "--variant 7 --exec-variant 304" "--variant 7 --exec-variant 308" "--variant 7 --exec-variant 124"; do
  echo "## $v" >> $O/lz4_corpus.txt
  timeout 120 $B --no-extra --data corpus --steps 5 --warmup 2 $v 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'], r['config']['decoder'][:30], r['config']['twopass_fallback_blocks'])" >> $O/lz4_corpus.txt 2>&1
done
stamp "snappy corpus: auto / parts"
for v in "" "--variant 7 --exec-variant 304"; do
  echo "## $v" >> $O/snappy_corpus.txt
  timeout 120 $B --no-extra --workload snappy_decompress --data corpus --steps 5 --warmup 2 $v 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'], r['config']['decoder'][:30])" >> $O/snappy_corpus.txt 2>&1
done

stamp "containers: defaults, then the prepared reader variants"
timeout 200 $B --section lz4frame > $O/containers_default.json 2> $O/containers_default.err
timeout 200 $B --section lz4frame --hadoop-variant 2 --snappyframed-variant 2 --lz4frame-variant 1 > $O/containers_new.json 2> $O/containers_new.err

stamp "zstd section: default / lit 8 / seq 8 / both / window 8192"
timeout 200 $B --section zstd > $O/zstd_default.json 2> $O/zstd_default.err
timeout 200 $B --section zstd --zstd-lit-items 8 > $O/zstd_lit8.json 2> $O/zstd_lit8.err
timeout 200 $B --section zstd --zstd-seq-items 8 > $O/zstd_seq8.json 2> $O/zstd_seq8.err
timeout 200 $B --section zstd --zstd-exec-window 8192 > $O/zstd_win8k.json 2> $O/zstd_win8k.err
stamp "zstd streams: default pass size / window 8192"
timeout 200 $B --section zstdstream > $O/zstdstream_default.json 2> $O/zstdstream_default.err
timeout 200 $B --section zstdstream --zstd-exec-window 8192 > $O/zstdstream_win8k.json 2> $O/zstdstream_win8k.err

stamp "encoders: lz4 / snappy on corpus and fragments, variant default vs 3"
for wl in lz4_compress snappy_compress; do
  for d in corpus fragments; do
    for v in "" "--compress-variant 3"; do
      echo "## $wl $d $v" >> $O/encoders.txt
      timeout 150 $B --no-extra --workload $wl --data $d --blocks 65536 --steps 3 --warmup 1 $v 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'])" >> $O/encoders.txt 2>&1
    done
  done
done

stamp "N = 2 path check on one device"
ACHIP_BENCH_SHARE_DEVICE=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --blocks 65536 --no-extra > $O/n2_pathcheck.json 2> $O/n2_pathcheck.err
tail -c 600 $O/n2_pathcheck.json | tee -a $O/timeline.txt

stamp "counter calibration"
timeout 150 bash tools/calibrate_write_size.sh > $O/calibration.txt 2>&1
mv gpurun_out/calib $O/calib 2>/dev/null

stamp "zstd corpus: per-kernel stats"
OUT=$O/ks_zstd; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python bench.py --no-cpu-baseline --no-sweep --section zstd > $OUT/log.txt 2>&1
F=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && cp "$F" $O/zstd_kernel_stats.csv
rm -rf $OUT
stamp "done"
