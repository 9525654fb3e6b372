mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 1200 python -m pytest tests/test_gpu_synthesis.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" > gpurun_out/t1.log
cat gpurun_out/t1.log
for v in "" world_class_amd/_variants/rows4.so world_class_amd/_variants/rows1.so ""; do
WC_LIB_PATH=$v python tools/microbench.py --stages cds --utts 64 --iters 5 2>&1 | grep -E "synthesis_pulses"
done
