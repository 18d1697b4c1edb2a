#!/bin/bash
# the host-pointer pipeline over chunk size x slots (x further options): tools/r06/hostsweep.sh "<chunkMiB> <opt>=<v> ..." ...   (three runs each)
mkdir -p gpurun_out/r06
for spec in "$@"; do
  set -- $spec
  ch=$1; shift
  for rep in 1 2 3; do echo -n "chunk $ch MiB $*: "; timeout 300 python tools/host_path_rate.py 0 $ch "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c30-; done
done | tee -a gpurun_out/r06/hostsweep.txt
