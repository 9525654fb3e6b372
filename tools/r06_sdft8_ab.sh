#!/bin/bash
# The eight-lane sliding band-pass step by step (WC_HARVEST_SDFT_LANES=9: rounds 3-5) against its block form (default) at small batches,
# and the block form kept beyond the threshold of 3072 wavefronts:  bash tools/r06_sdft8_ab.sh > gpurun_out/sdft8_ab.txt 2>&1
python -m pytest tests/test_gpu_harvest.py -q -x -k "eight_lanes" 2>&1 | tail -3
for v in "WC_HARVEST_SDFT_LANES=9" "WC_X=0" "WC_HARVEST_SDFT8_MAX=8000" "WC_HARVEST_SDFT8_MAX=16000" "WC_X=1"; do
  echo "== $v"
  env $v LAT_N=1,2,4,8,16,32 python tools/latency_probe.py 2>&1 | grep utterances
done
