#!/bin/bash
# A third parity sweep with seeds of its own (round 5: the same shapes as tools/final_sweep.sh, other signals; 22.05 kHz instead of 96 kHz):
#   bash tools/third_sweep.sh > gpurun_out/sweep3.txt 2>&1
S="timeout 1700 python tests/parity_sweep.py"
$S --n 300 --first-seed 1400000 --fs 48000 --seconds 10 --ragged
$S --n 200 --first-seed 1410000 --fs 16000 --seconds 8 --ragged --floor 40
$S --n 100 --first-seed 1420000 --fs 44100 --seconds 5 --ragged --frame-period 1
$S --n 100 --first-seed 1430000 --fs 24000 --seconds 4 --ragged --frame-period 1
$S --n 100 --first-seed 1440000 --zoo --fs 16000 --seconds 3
$S --n 60 --first-seed 1450000 --zoo --dither 1e-3 --fs 48000 --seconds 2
$S --n 40 --first-seed 1460000 --fs 22050 --seconds 3 --ragged
timeout 900 python tests/stage_sweep.py --n 100 --first-seed 1470000
