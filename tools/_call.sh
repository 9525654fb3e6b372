O=gpurun_out/r05_p20; mkdir -p $O
S="timeout 2500 python tests/parity_sweep.py"
( $S --n 60 --first-seed 1500000 --zoo --fs 48000 --seconds 3
  $S --n 60 --first-seed 1510000 --zoo --fs 44100 --seconds 3
  $S --n 60 --first-seed 1520000 --zoo --fs 24000 --seconds 3 --frame-period 1
  $S --n 60 --first-seed 1530000 --zoo --fs 22050 --seconds 3
  $S --n 40 --first-seed 1550000 --zoo --fs 96000 --seconds 2
  $S --n 60 --first-seed 1560000 --zoo --fs 16000 --seconds 6 --floor 40
  $S --n 60 --first-seed 1570000 --zoo --fs 32000 --seconds 3 ) > $O/zoo_rates_port.txt 2>&1
grep -v amdgpu $O/zoo_rates_port.txt | grep "^fs"
python -m pytest tests/test_gpu_sweeps.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
