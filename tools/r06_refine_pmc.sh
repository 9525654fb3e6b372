# counter passes over the Harvest stage alone, one refinement layout per pass:  bash tools/r06_refine_pmc.sh "group packed"
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/r06_refine_pmc
mkdir -p $OUT
cd /tmp
for mode in $1; do
  for pass in "valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES SQ_WAVES"; do
    set -- $pass
    name=$1; shift
    ( cd $REPO && WC_HARVEST_REFINE=$mode rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/${mode}_$name" -o p -- python tools/microbench.py --stages h --utts 64 --iters 1 > "$OUT/${mode}_$name.txt" 2> "$OUT/${mode}_$name.err" )
  done
  echo "== $mode"
  ( cd $REPO && python tools/pmc_sq.py "$OUT/${mode}_valu" "$OUT/${mode}_f64" | grep -A16 "hv_refine" )
done
find "$OUT" -name "*.csv" -size +5M -delete
