mkdir -p gpurun_out
{
for rep in 1 2; do
for v in plain low; do
  echo "== WC_PIPELINE_BACK=$v"
  PIN_IN=1 WC_PIPELINE_BACK=$v python tools/host_frontend_probe.py 2>&1 | grep -E "^run" | tr "\n" " "; echo
done
done
echo "== timing, plain"
PIN_IN=1 WC_PIPELINE_BACK=plain WC_PIPELINE_TIMING=1 python tools/host_frontend_probe.py 2>&1 | grep -E "group [0-9] \(|returned" | tail -7
echo "== timing, low"
PIN_IN=1 WC_PIPELINE_BACK=low WC_PIPELINE_TIMING=1 python tools/host_frontend_probe.py 2>&1 | grep -E "group [0-9] \(|returned|landed" | tail -12
} > gpurun_out/back_low.txt 2>&1
cat gpurun_out/back_low.txt
