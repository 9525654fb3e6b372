python -m pytest tests/test_gpu_harvest.py tests/test_gpu_sweeps.py tests/test_gpu_pipeline.py tests/test_gpu_stream.py tests/test_gpu_robustness.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
python tools/zoo_diag.py 96000 1550002 2 5 24000 1520002 3 1 2>&1 | grep -E "gpu\(sdft\) vs ref"
python tests/parity_sweep.py --n 40 --first-seed 1710000 --fs 16000 --seconds 4 --ragged --zoo 2>&1 | grep "^fs"
