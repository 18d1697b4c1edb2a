#!/bin/bash
# Round 3, call 16: the stream writer's private block buffer on the GPU (test_gpu_zstd_stream.py), then the differential fuzz of every encoder
# variant -- the round's new window encoders first -- against the oracle: 4 seeds x 1500 inputs per codec.  -> gpurun_out/r03c16/
export TMPDIR=/tmp
O=gpurun_out/r03c16
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_zstd_stream.py tests/test_gpu_zstd.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for seed in 21 22 23 24; do
  timeout 900 python tools/fuzz_encoders.py 1500 $seed lz4,snappy,zstd 2>&1 | grep -v "^$" | tee -a $O/fuzz_encoders.txt | tail -12
done
grep -c "0 mismatches" $O/fuzz_encoders.txt; grep "TOTAL" $O/fuzz_encoders.txt
