#!/bin/bash
# A second parity sweep with seeds of its own (round 4: the same shapes as tools/final_sweep.sh, other signals; 22.05 kHz instead of 96 kHz):
#   bash tools/second_sweep.sh > gpurun_out/sweep2.txt 2>&1
S="timeout 1700 python tests/parity_sweep.py"
$S --n 300 --first-seed 1300000 --fs 48000 --seconds 10 --ragged
$S --n 200 --first-seed 1310000 --fs 16000 --seconds 8 --ragged --floor 40
$S --n 100 --first-seed 1320000 --fs 44100 --seconds 5 --ragged --frame-period 1
$S --n 100 --first-seed 1330000 --fs 24000 --seconds 4 --ragged --frame-period 1
$S --n 100 --first-seed 1340000 --zoo --fs 16000 --seconds 3
$S --n 60 --first-seed 1350000 --zoo --dither 1e-3 --fs 48000 --seconds 2
$S --n 40 --first-seed 1360000 --fs 22050 --seconds 3 --ragged
timeout 900 python tests/stage_sweep.py --n 100 --first-seed 1370000
