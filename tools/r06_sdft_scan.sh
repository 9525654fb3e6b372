# one-lane sliding band-pass by batch size, detectors at every step (7) against deferred (default):  bash tools/r06_sdft_scan.sh
for n in 16 24 32 33 40 48 64; do
  for v in WC_HARVEST_SDFT_LANES=7 WC_HARVEST_SDFT_LANES=1; do
    echo -n "utts $n $v: "; env $v python tools/microbench.py --stages h --utts $n --iters 5 2>&1 | grep "harvest_bandpass"
  done
done
