mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^E|assert|passed|failed" | head -20 > gpurun_out/t.txt
python tools/latency_probe.py 2>&1 | grep utterances >> gpurun_out/t.txt
cat gpurun_out/t.txt
