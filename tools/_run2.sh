mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/tall.log
cat gpurun_out/tall.log
