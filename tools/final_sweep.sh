#!/bin/bash
# Parity sweep of the final build of a round against the CPU oracle, far beyond the committed fixtures (GPU box; ~15 min):
#   bash tools/final_sweep.sh > gpurun_out/sweep.txt 2>&1
S="timeout 1700 python tests/parity_sweep.py"
$S --n 300 --first-seed 300000 --fs 48000 --seconds 10 --ragged
$S --n 300 --first-seed 310000 --fs 16000 --seconds 8 --ragged --floor 40
$S --n 100 --first-seed 320000 --fs 44100 --seconds 5 --ragged --frame-period 1
$S --n 100 --first-seed 330000 --fs 24000 --seconds 4 --ragged --frame-period 1
$S --n 100 --first-seed 340000 --zoo --fs 16000 --seconds 3
$S --n 60 --first-seed 350000 --zoo --dither 1e-3 --fs 48000 --seconds 2
$S --n 40 --first-seed 360000 --fs 96000 --seconds 3 --ragged
timeout 900 python tests/stage_sweep.py --n 100 --first-seed 370000
