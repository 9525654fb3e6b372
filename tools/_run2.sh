mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>gpurun_out/bench.err
tail -3 gpurun_out/bench.err; cat gpurun_out/bench.log
