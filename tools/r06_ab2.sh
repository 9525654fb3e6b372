for mode in group group8 group group8; do
  echo "== WC_HARVEST_REFINE=$mode"
  WC_HARVEST_REFINE=$mode python tools/microbench.py --utts 64 --iters 5 --stages h 2>&1 | grep "refine"
done
