python -m pytest tests/test_gpu_harvest.py tests/test_gpu_blocks.py -x -q 2>&1 | tail -3
for mode in blocks default blocks default; do
  echo "== WC_HARVEST_RAW=$mode"
  WC_HARVEST_RAW=$mode python tools/microbench.py --utts 64 --iters 5 --stages h 2>&1 | grep "raw"
done
