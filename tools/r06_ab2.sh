python -m pytest tests/test_gpu_cheaptrick.py -x -q 2>&1 | tail -2
for lib in "" world_class_amd/_variants/ct_before.so "" world_class_amd/_variants/ct_before.so; do
  echo "== lib=$lib"
  WC_LIB_PATH=$lib python tools/microbench.py --stages c --utts 256 --iters 5 2>&1 | grep cheaptrick
done
