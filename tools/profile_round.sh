#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_* and profiles/pmc_traffic.json, run on the GPU box:
#   bash tools/profile_round.sh r03_a
# Counter passes are separate runs with --kernel-trace only (never combined with other trace domains); outputs go to
# gpurun_out/<tag>/, which gpurun merges back; copy what is to be kept into profiles/.
set -u
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-serialised"
REPO=$PWD
HASH=$(python -c "import world_class_amd as w; print(w.lib().wc_build_hash().decode())")
cd /tmp
# 1. kernel-trace statistics of the bench command and of BASELINE config 3 (CheapTrick alone)
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o p -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-serialised > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || true
( cd $REPO && rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_config3" -o p -- python tools/microbench.py --stages c --utts 256 --iters 3 > "$OUT/stats_config3.txt" 2> "$OUT/stats_config3.err" ) || true
# 2. counter passes over the bench command
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "valu SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
  set -- $pass
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o p -- python $REPO/bench.py $BENCH_ARGS > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || true
done
# 3. config 3 alone: HBM bytes and FP64 operations of the CheapTrick kernel
for pass in "c3fetch FETCH_SIZE" "c3write WRITE_SIZE" "c3f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "c3valu SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  set -- $pass
  name=$1; shift
  ( cd $REPO && rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o p -- python tools/microbench.py --stages c --utts 256 --iters 1 > "$OUT/pmc_$name.txt" 2> "$OUT/pmc_$name.err" ) || true
done
# 4. calibration of FETCH_SIZE on streaming reads of known size (8 and 16 bytes per lane)
( cd $REPO && rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/pmc_calib" -o p -- python tools/fetch_calibrate.py > "$OUT/pmc_calib.txt" 2> "$OUT/pmc_calib.err" ) || true
cd "$REPO"
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" --valu-dir "$OUT/pmc_valu" --f64-dir "$OUT/pmc_f64" --steps 3 --build-hash "$HASH" \
  --calib "$OUT/pmc_calib" --config3-f64-dir "$OUT/pmc_c3f64" --config3-valu-dir "$OUT/pmc_c3valu" -o "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1
python tools/pmc_traffic.py "$OUT/pmc_c3fetch" "$OUT/pmc_c3write" --steps 2 --calib "$OUT/pmc_calib" -o "$OUT/pmc_traffic_config3.json" > "$OUT/pmc_traffic_config3.log" 2>&1
python tools/pmc_sq.py "$OUT/pmc_valu" > "$OUT/sq_counters.txt" 2>&1
python tools/pmc_sq.py "$OUT/pmc_c3valu" > "$OUT/sq_counters_config3.txt" 2>&1
python tools/isa_mix.py > "$OUT/isa_mix.txt" 2>&1
python tools/kernel_resources.py > "$OUT/kernel_resources.txt" 2>&1
# keep the merge small: the per-dispatch csv files of the counter passes are large
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
ls -la "$OUT"
