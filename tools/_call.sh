mkdir -p gpurun_out
python -m pytest tests/test_gpu_harvest.py tests/test_gpu_synthesis.py tests/test_gpu_multirank.py tests/test_abi_and_boundary.py -m gpu -x -q 2>&1 | grep -E "^E|assert|passed|failed" | head -20 > gpurun_out/t.txt
cat gpurun_out/t.txt
