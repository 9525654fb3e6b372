for v in 2 0 1 2 0 1; do
  echo "== WC_PIPELINE_PRE_LANE=$v"
  WC_PIPELINE_PRE_LANE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-serialised 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
