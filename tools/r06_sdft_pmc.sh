# SQ counters of the one-lane sliding band-pass, detectors at every step (7) against deferred (1):
#   bash tools/r06_sdft_pmc.sh "7 1" > gpurun_out/sdft_pmc.txt 2>&1
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/r06_sdft_pmc
mkdir -p $OUT
for mode in $1; do
  for pass in "valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
    set -- $pass
    name=$1; shift
    ( cd $REPO && WC_HARVEST_SDFT_LANES=$mode rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/${mode}_$name" -o p -- python tools/microbench.py --stages h --utts 64 --iters 1 > "$OUT/${mode}_$name.txt" 2> "$OUT/${mode}_$name.err" )
  done
  echo "== WC_HARVEST_SDFT_LANES=$mode"
  ( cd $REPO && python tools/pmc_sq.py "$OUT/${mode}_valu" "$OUT/${mode}_f64" | grep -A20 "hv_bandpass_sdft_kernel" | head -22 )
done
find "$OUT" -name "*.csv" -size +5M -delete
