mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 1200 python -m pytest tests/test_gpu_cheaptrick.py tests/test_gpu_d4c.py tests/test_gpu_pipeline.py tests/test_gpu_synthesis.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t1.log
cat gpurun_out/t1.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b1.log 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/b1.log').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms'])
PY
python tools/latency_probe.py 2>&1 | grep utter
