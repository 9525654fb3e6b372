# A/B of the refinement kernels on the GPU box: bit-identity tests, then stage timings with each layout
mkdir -p gpurun_out
python -m pytest tests/test_gpu_harvest.py -x -q 2>&1 | tail -3
for mode in group group2 packed; do
  echo "== WC_HARVEST_REFINE=$mode"
  WC_HARVEST_REFINE=$mode python tools/microbench.py --utts 64 --iters 3 --stages h 2>&1 | grep "refine\|wall"
done
if [ -f world_class_amd/_variants/rqprof.so ]; then WC_LIB_PATH=world_class_amd/_variants/rqprof.so python tools/microbench.py --stages h --utts 64 --iters 1 2>&1 | tail -18 | head -12; fi
