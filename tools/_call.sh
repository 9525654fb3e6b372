mkdir -p gpurun_out
{
for fs in 16000 24000; do
for v in "WC_PIPELINE_UNCHAIN_BELOW=0" "A=1" "WC_PIPELINE_UNCHAIN_BELOW=0" "A=1"; do echo "== fs $fs $v"; env LAT_FS=$fs LAT_N=8,16,32,48 $v python tools/latency_probe.py 2>&1 | grep utterances; done
done
} > gpurun_out/side16b.txt 2>&1
cat gpurun_out/side16b.txt
