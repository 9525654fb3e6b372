python -m pytest tests/test_gpu_harvest.py -x -q 2>&1 | tail -3
for mode in default chunks default chunks; do
  echo "== WC_HARVEST_DECIMATE=$mode"
  WC_HARVEST_DECIMATE=$mode python tools/microbench.py --utts 64 --iters 5 --stages h 2>&1 | grep "decimate"
  WC_HARVEST_DECIMATE=$mode python tools/microbench.py --utts 32 --iters 5 --stages h 2>&1 | grep "decimate"
  WC_HARVEST_DECIMATE=$mode python tools/microbench.py --utts 1 --iters 5 --stages h 2>&1 | grep "decimate"
done
