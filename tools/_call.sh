mkdir -p gpurun_out/r05_p2
python -m pytest tests/test_gpu_synthesis.py tests/test_gpu_batch_host.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r05_p2/tests.log 2>&1; tail -3 gpurun_out/r05_p2/tests.log
run() { # name, env...
  name=$1; shift
  echo "== $name: $*" >> gpurun_out/r05_p2/sweep.txt
  env PIN_IN=1 "$@" python tools/host_frontend_probe.py 2>&1 | grep "^run" >> gpurun_out/r05_p2/sweep.txt
}
for syn in 0 1; do
 for cm in 0 4; do
  run base WC_PIPELINE_SYN_STREAMS=$syn WC_PIPELINE_CHAIN_MIN=$cm
  run s5 WC_PIPELINE_SYN_STREAMS=$syn WC_PIPELINE_CHAIN_MIN=$cm WC_PIPELINE_HOST_SPLITS=5,8,12,18,25
  run s6 WC_PIPELINE_SYN_STREAMS=$syn WC_PIPELINE_CHAIN_MIN=$cm WC_PIPELINE_HOST_SPLITS=3,5,8,12,17,23
  run s7 WC_PIPELINE_SYN_STREAMS=$syn WC_PIPELINE_CHAIN_MIN=$cm WC_PIPELINE_HOST_SPLITS=3,4,6,9,13,18,22
  run s4 WC_PIPELINE_SYN_STREAMS=$syn WC_PIPELINE_CHAIN_MIN=$cm WC_PIPELINE_HOST_SPLITS=6,10,15,22
 done
done
WC_PIPELINE_TIMING=1 PIN_IN=1 WC_PIPELINE_SYN_STREAMS=1 python tools/host_frontend_probe.py > gpurun_out/r05_p2/timing_syn1.txt 2>&1
WC_PIPELINE_TIMING=1 PIN_IN=1 WC_PIPELINE_SYN_STREAMS=1 WC_PIPELINE_HOST_SPLITS=3,5,8,12,17,23 python tools/host_frontend_probe.py > gpurun_out/r05_p2/timing_syn1_s6.txt 2>&1
cat gpurun_out/r05_p2/sweep.txt
