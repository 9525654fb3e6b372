O=gpurun_out/r05_p14; mkdir -p $O
python tools/impulse_diag2.py 1440023 > $O/diag2_1440023.txt 2>&1; grep -v amdgpu $O/diag2_1440023.txt | head -16
WC_HARVEST_TIES=ignore python tools/impulse_diag.py > $O/impulse_diag_ignore.txt 2>&1; grep -v amdgpu $O/impulse_diag_ignore.txt | grep -E "^default|^pipeline 1440023" | head -20
python tools/microbench.py --stages h --utts 64 --iters 3 > $O/mb_h.txt 2>&1; tail -7 $O/mb_h.txt
python -m pytest tests/test_gpu_harvest.py tests/test_gpu_sweeps.py tests/test_gpu_pipeline.py tests/test_gpu_robustness.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
