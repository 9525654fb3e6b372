mkdir -p gpurun_out/r05_p4
O=gpurun_out/r05_p4
hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate tools/issue_rate.hip && /tmp/issue_rate $O/issue_rates.json > $O/issue_rates.txt 2>&1
cat $O/issue_rates.txt
python tools/microbench.py --utts 64 --iters 3 > $O/mb_base.txt 2>&1
WC_SYN_OLA=atomic python tools/microbench.py --utts 64 --iters 3 > $O/mb_atomic.txt 2>&1
WC_LIB_PATH=world_class_amd/_variants/alias.so python tools/microbench.py --utts 64 --iters 3 > $O/mb_alias.txt 2>&1
WC_LIB_PATH=world_class_amd/_variants/alias.so WC_SYN_OLA=atomic python tools/microbench.py --utts 64 --iters 3 > $O/mb_alias_atomic.txt 2>&1
tail -14 $O/mb_base.txt; grep -E "d4c_frames|synthesis_pulses|wall" $O/mb_atomic.txt $O/mb_alias.txt $O/mb_alias_atomic.txt
WC_SYN_OLA=atomic python -m pytest tests/test_gpu_synthesis.py tests/test_gpu_pipeline.py -m gpu -q -k "golden or oracle or ragged" > $O/tests_atomic.log 2>&1; grep -E "passed|failed" $O/tests_atomic.log
for ola in rows atomic; do
 WC_SYN_OLA=$ola python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$ola.json 2> $O/bench_$ola.err
 python -c "
import json; d=json.load(open('$O/bench_$ola.json')); print('$ola', d['ms_per_step'], d['value'])"
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
python -c "
import json; d=json.load(open('$O/bench_full.json')); print(d['ms_per_step']); print({k:(round(v['ms'],1) if 'ms' in v else v) for k,v in d['with_transfers'].items()}); print({k:(round(v['ms'],2) if 'ms' in v else '') for k,v in d['stages'].items()})"
