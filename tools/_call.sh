O=gpurun_out/r05_p6; mkdir -p $O
for v in default bandpf0 bandpf192 bandpf768 synpf200 synpf60; do
  if [ $v = default ]; then unset WC_LIB_PATH; else export WC_LIB_PATH=world_class_amd/_variants/$v.so; fi
  python tools/microbench.py --utts 64 --iters 4 > $O/mb_$v.txt 2>&1
  echo "$v: $(grep -E 'd4c_bands|synthesis_pulses' $O/mb_$v.txt | tr '\n' ' ')"
done
unset WC_LIB_PATH
python -m pytest tests/test_gpu_d4c.py tests/test_gpu_synthesis.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
WC_LIB_PATH=world_class_amd/_variants/synpf200.so python -m pytest tests/test_gpu_synthesis.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
for v in default bandpf0; do
  if [ $v = default ]; then unset WC_LIB_PATH; else export WC_LIB_PATH=world_class_amd/_variants/$v.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-serialised > $O/bench_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', round(d['ms_per_step'],2), round(d['value']))"
done
