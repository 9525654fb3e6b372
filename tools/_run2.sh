mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 900 python -m pytest tests/test_gpu_d4c.py -x -q 2>&1 | tail -3 > gpurun_out/t2.log
timeout 300 python tools/microbench.py --stages cd --utts 64 --iters 3 >> gpurun_out/d4.log 2>&1
cat gpurun_out/t2.log gpurun_out/d4.log
