# Full evidence pass of a build on the GPU box: GPU tests, bench.py in the driver-run shape, tools/profile_round.sh (all rocprofv3 passes),
# stage-by-stage timings and the small-batch latency probe; outputs under gpurun_out/, copied into profiles/ by hand:  bash tools/evidence_round.sh TAG
TAG=$1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest.log 2>&1; grep -E 'passed|failed' gpurun_out/${TAG}_gputest.log | tail -3
# (the counter passes first: bench.py reads profiles/pmc_traffic.json of THIS build for roofline.traffic / issue_frac and refuses another build's)
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
cp gpurun_out/$TAG/pmc_traffic.json profiles/pmc_traffic.json
# the issue-rate calibration of THIS box (bench.py prices issue_frac on profiles/issue_rates.json)
hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate tools/issue_rate.hip && /tmp/issue_rate gpurun_out/${TAG}_issue_rates.json > gpurun_out/${TAG}_issue_rates.txt 2>&1 && cp gpurun_out/${TAG}_issue_rates.json profiles/issue_rates.json
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python tools/microbench.py --utts 64 --iters 3 > gpurun_out/${TAG}_mb64.txt 2>&1
python tools/microbench.py --utts 16 --iters 3 > gpurun_out/${TAG}_mb16.txt 2>&1
python tools/microbench.py --utts 1 --iters 5 > gpurun_out/${TAG}_mb1.txt 2>&1
python tools/latency_probe.py > gpurun_out/${TAG}_latency.txt 2>&1
PIN_IN=1 WC_PIPELINE_TIMING=1 python tools/host_frontend_probe.py > gpurun_out/${TAG}_host_frontend_timing.txt 2>&1
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('fp64_vector')); print({k:(round(v['ms'],1) if 'ms' in v else v) for k,v in d['with_transfers'].items()}); print(d['stages']['cheaptrick_config3'])"
cat gpurun_out/${TAG}_latency.txt
