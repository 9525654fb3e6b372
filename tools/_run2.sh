mkdir -p gpurun_out; rm -f gpurun_out/*.log
( time timeout 1200 python bench.py ) > gpurun_out/bench.log 2>gpurun_out/bench.err
tail -5 gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"vwt",d.get("value_with_transfers"))
print(json.dumps(d["roofline"]["all_kernels_ms"]))
print(json.dumps(d["stages"],indent=0)[:3000])
print(json.dumps(d["with_transfers"],indent=0)[:1500])
print(json.dumps(d.get("cpu_baseline"))[:400])
PY
