#!/bin/bash
# SQ counter passes over any command (development aid), on the GPU box:
#   bash tools/pmc_micro.sh TAG python tools/microbench.py --stages cd --utts 64 --iters 2
# Counter passes are separate runs with --kernel-trace only; summaries land in gpurun_out/TAG/.
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
            "SQ_WAVE_CYCLES SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  ( cd $REPO && rocprofv3 --kernel-trace --pmc $pass -f csv -d "$OUT/pmc_$i" -o p -- "$@" > "$OUT/pmc_$i.out" 2> "$OUT/pmc_$i.err" ) || true
done
cd "$REPO"
python tools/pmc_sq.py "$OUT/pmc_1" "$OUT/pmc_2" "$OUT/pmc_3" > "$OUT/sq_counters.txt" 2>&1
find "$OUT" -name "*.csv" -size +5M -delete
tail -n +1 "$OUT/sq_counters.txt" | head -150
