mkdir -p gpurun_out
{
for v in x h c x h c; do echo "== WC_PIPELINE_SIDE=$v"; LAT_N=56,64 WC_PIPELINE_SIDE=$v python tools/latency_probe.py 2>&1 | grep utterances; done
} > gpurun_out/side.txt 2>&1
cat gpurun_out/side.txt
