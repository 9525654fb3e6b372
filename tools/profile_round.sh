#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_* and profiles/pmc_traffic.json, run on the GPU box:
#   bash tools/profile_round.sh r02_a
# Counter passes are separate runs with --kernel-trace only (never combined with other trace domains); outputs go to
# gpurun_out/<tag>/, which gpurun merges back; copy what is to be kept into profiles/.
set -u
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras"
OLDPWD_REPO=$PWD
cd /tmp
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o p -- python $OLDPWD_REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || true
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "valu SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
  set -- $pass
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o p -- python $OLDPWD_REPO/bench.py $BENCH_ARGS > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || true
done
cd "$OLDPWD_REPO"
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" --valu-dir "$OUT/pmc_valu" --f64-dir "$OUT/pmc_f64" --steps 3 -o "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1
python tools/pmc_sq.py "$OUT/pmc_valu" > "$OUT/sq_counters.txt" 2>&1
# keep the merge small: the per-dispatch csv files of the counter passes are large
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
ls -la "$OUT"
