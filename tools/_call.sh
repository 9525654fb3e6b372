mkdir -p gpurun_out
python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k ties 2>&1 | grep -E "^E|assert|passed|failed" | head -20 > gpurun_out/stream_ties.txt
cat gpurun_out/stream_ties.txt
