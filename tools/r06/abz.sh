#!/bin/bash
# several builds of the library on ONE box, the Zstd section of bench.py: tools/r06/abz.sh "<kinds>" <lib> ...   ("new" = the tree's library)
KINDS=$1; shift
L=aircompressor_amd/libaircompressor_hip.so
cp $L /tmp/lib_new.so
for rep in 1 2; do for lib in "$@"; do
  if [ $lib = new ]; then cp /tmp/lib_new.so $L; else cp tools/r06/lib_$lib.so $L; fi
  timeout 600 python bench.py --section zstd --zstd-kinds $KINDS --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$lib', {k:(v['decompress_GiBps'], v['java_frames_decompress_GiBps']) for k,v in r.items()})"
done; done
cp /tmp/lib_new.so $L
