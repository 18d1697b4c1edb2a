#!/bin/bash
# several builds of the library on ONE box: tools/r06/abc.sh "<bench args>" <lib>[:option=value] ...   ("new" = the tree's library)
ARGS=$1; shift
L=aircompressor_amd/libaircompressor_hip.so
cp $L /tmp/lib_new.so
for rep in 1 2; do for spec in "$@"; do
  lib=${spec%%:*}; opt=""; [ "$spec" != "$lib" ] && opt="--option ${spec#*:}"
  if [ $lib = new ]; then cp /tmp/lib_new.so $L; else cp tools/r06/lib_$lib.so $L; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extra --no-legs --no-host-facing --no-sweep --steps 3 --warmup 1 $ARGS $opt 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$spec', r['value'], 'GiB/s  kernel ms', r['roofline'].get('kernel_ms_avg'))"
done; done
cp /tmp/lib_new.so $L
