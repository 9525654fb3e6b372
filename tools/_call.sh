mkdir -p gpurun_out
{
for fs in 16000 24000; do
for v in "A=1" "WC_PIPELINE_UNCHAIN_BELOW=100000" "WC_PIPELINE_SIDE=c" "WC_PIPELINE_SIDE=h" "A=1" "WC_PIPELINE_UNCHAIN_BELOW=100000" "WC_PIPELINE_SIDE=c"; do echo "== fs $fs $v"; env LAT_FS=$fs LAT_N=48,64 $v python tools/latency_probe.py 2>&1 | grep utterances; done
done
} > gpurun_out/side16.txt 2>&1
cat gpurun_out/side16.txt
