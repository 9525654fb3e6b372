mkdir -p gpurun_out
{
for v in 50 35 40 45 55 60 50 40 45; do echo "== WC_PIPELINE_FIRST_SHARE=$v"; LAT_N=16,32,48,64 WC_PIPELINE_FIRST_SHARE=$v python tools/latency_probe.py 2>&1 | grep utterances | tr "\n" " "; echo; done
} > gpurun_out/share.txt 2>&1
cat gpurun_out/share.txt
