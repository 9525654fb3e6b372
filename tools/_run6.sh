python -m pytest tests/test_gpu_harvest.py tests/test_gpu_robustness.py -x -q 2>&1 | grep -E "passed|failed|Error"
for v in ring noring; do
  if [ $v = noring ]; then export WC_LIB_PATH=world_class_amd/_variants/noring.so; fi
  echo $v; python tools/microbench.py --stages h --utts 64 --iters 5 | tail -7
done
unset WC_LIB_PATH
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/gpurun_out/bpw -o p -- python $R/tools/microbench.py --stages h --utts 64 --iters 1 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/bpw/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'WRITE_SIZE':
        acc[r['Kernel_Name'][:50]] += float(r['Counter_Value']); n[r['Kernel_Name'][:50]] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:6]:
    print("%-52s launches %3d  WRITE_SIZE %.3f GB total" % (k, n[k], v * 1024 / 1e9))
PY
