mkdir -p gpurun_out
{
for v in "A=1" "WC_PIPELINE_SHARE=1" "WC_PIPELINE_SHARE=2" "WC_PIPELINE_SHARE=3" "GPU_MAX_HW_QUEUES=2" "A=1" "WC_PIPELINE_SHARE=1" "WC_PIPELINE_SHARE=2" "WC_PIPELINE_SHARE=3" "GPU_MAX_HW_QUEUES=2"; do echo "== $v"; env LAT_N=56,64 $v python tools/latency_probe.py 2>&1 | grep utterances | tr "\n" " "; echo; done
} > gpurun_out/share2.txt 2>&1
cat gpurun_out/share2.txt
