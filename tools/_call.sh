O=gpurun_out/r05_p22; mkdir -p $O
S="timeout 2500 python tests/parity_sweep.py"
( $S --n 60 --first-seed 1600000 --fs 8000 --seconds 5 --ragged
  $S --n 60 --first-seed 1610000 --fs 11025 --seconds 5 --ragged
  $S --n 60 --first-seed 1620000 --fs 32000 --seconds 5 --ragged --floor 40
  $S --n 40 --first-seed 1630000 --fs 96000 --seconds 3 --ragged
  $S --n 40 --first-seed 1640000 --zoo --fs 48000 --seconds 2 --floor 40 --frame-period 1
  $S --n 60 --first-seed 1650000 --zoo --fs 16000 --seconds 3 --frame-period 1
  $S --n 60 --first-seed 1660000 --zoo --fs 24000 --seconds 3 --floor 40
  $S --n 100 --first-seed 1670000 --zoo --fs 16000 --seconds 0.6 --ragged
  $S --n 60 --first-seed 1680000 --zoo --fs 48000 --seconds 0.5 --ragged
  $S --n 60 --first-seed 1690000 --zoo --dither 1e-4 --fs 44100 --seconds 2 ) > $O/sweep4.txt 2>&1
timeout 1500 python tests/stage_sweep.py --n 200 --first-seed 1700000 >> $O/sweep4.txt 2>&1
grep -v amdgpu $O/sweep4.txt | tail -60 | cut -c1-200
