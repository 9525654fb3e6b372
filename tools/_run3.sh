mkdir -p gpurun_out
python tools/pcie_probe.py > gpurun_out/pcie.txt 2>&1
python tools/microbench.py --utts 64 --iters 3 > gpurun_out/mb64.txt 2>&1
python tools/microbench.py --utts 1 --iters 5 > gpurun_out/mb1.txt 2>&1
python tools/latency_probe.py > gpurun_out/latency.txt 2>&1
WC_LIB_PATH=world_class_amd/_variants/d4trace.so WC_D4C_TRACE=/tmp/t.bin python tools/microbench.py --stages cd --utts 8 --iters 1 > /dev/null 2>&1
python tools/d4c_trace.py /tmp/t.bin > gpurun_out/d4trace.txt 2>&1
WC_LIB_PATH=world_class_amd/_variants/syntrace.so WC_SYN_TRACE_FILE=/tmp/s.bin python tools/microbench.py --stages cds --utts 64 --iters 1 > /dev/null 2>&1
python tools/syn_trace.py /tmp/s.bin > gpurun_out/syntrace.txt 2>&1
cat gpurun_out/pcie.txt gpurun_out/mb64.txt gpurun_out/mb1.txt gpurun_out/latency.txt gpurun_out/d4trace.txt gpurun_out/syntrace.txt
