mkdir -p gpurun_out
python -m pytest tests/test_gpu_sweeps.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | grep -E "^E|assert|passed|failed" | head -20 > gpurun_out/t.txt
cat gpurun_out/t.txt
