mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" > gpurun_out/tall.log
cat gpurun_out/tall.log
python tools/latency_probe.py 2>&1 | grep utter
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/b1.log 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/b1.log').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms'])
print({k:v['ms'] for k,v in d['with_transfers'].items()})
print(d['cpu_baseline'])
print(d['stages']['dropin_single_utterance']['steady'])
PY
