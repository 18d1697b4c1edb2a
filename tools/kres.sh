#!/bin/bash
# per-kernel register / LDS / occupancy report of one source file (no GPU needed): tools/kres.sh <file.hip>
cd "$(dirname "$0")/../aircompressor_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -x hip -c "$1" -o /dev/null --cuda-device-only -Rpass-analysis=kernel-resource-usage 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Occupancy|LDS Size|ScratchSize" | sed -e 's/^.*remark: [^ ]* //' | paste - - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
