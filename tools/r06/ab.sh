#!/bin/bash
# A/B of two builds of the library on ONE box: tools/r06/ab.sh <workload> <data> [blocks] -- old (tools/r06/lib_old.so), new, old, new
WL=$1; DATA=$2; BL=${3:-65536}
L=aircompressor_amd/libaircompressor_hip.so
cp $L /tmp/lib_new.so
for rep in 1 2; do for which in old new; do
  if [ $which = old ]; then cp tools/r06/lib_old.so $L; else cp /tmp/lib_new.so $L; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extra --no-legs --no-host-facing --no-sweep --steps 3 --warmup 1 --workload $WL --data $DATA --blocks $BL 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$which $WL $DATA', r['value'], 'GiB/s  kernel ms', r['roofline'].get('kernel_ms_avg'))"
done; done
cp /tmp/lib_new.so $L
