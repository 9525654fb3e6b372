mkdir -p gpurun_out; rm -f gpurun_out/*.log
WC_LIB_PATH=world_class_amd/_variants/synnoat.so python tools/microbench.py --stages cds --utts 64 --iters 3 > gpurun_out/tr.log 2>&1
tail -8 gpurun_out/tr.log
