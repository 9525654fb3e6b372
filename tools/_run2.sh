mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t1.log
cat gpurun_out/t1.log
for m in 2 1; do
WC_PIPELINE_COPY_STREAMS=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b$m.log 2>&1; python - $m <<'PY'
import json,sys
d=json.loads(open('gpurun_out/b%s.log'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], {k:v['ms'] for k,v in d['with_transfers'].items()})
PY
done
python tools/latency_probe.py 2>&1 | grep utter
