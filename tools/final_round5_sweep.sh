#!/bin/bash
# The sweep on the LAST build of round 5 (the shapes of tools/third_sweep.sh with seeds of its own, plus the second set of kinds):
#   bash tools/final_round5_sweep.sh > gpurun_out/sweep7.txt 2>&1
S="timeout 1700 python tests/parity_sweep.py"
$S --n 300 --first-seed 1900000 --fs 48000 --seconds 10 --ragged
$S --n 200 --first-seed 1910000 --fs 16000 --seconds 8 --ragged --floor 40
$S --n 100 --first-seed 1920000 --fs 44100 --seconds 5 --ragged --frame-period 1
$S --n 100 --first-seed 1930000 --fs 24000 --seconds 4 --ragged --frame-period 1
$S --n 100 --first-seed 1940000 --zoo --fs 16000 --seconds 3
$S --n 60 --first-seed 1950000 --zoo --dither 1e-3 --fs 48000 --seconds 2
$S --n 40 --first-seed 1960000 --fs 32000 --seconds 3 --ragged
$S --n 60 --first-seed 1980000 --zoo2 --dither 1e-3 --fs 48000 --seconds 2
$S --n 60 --first-seed 1990000 --zoo2 --dither 1e-4 --fs 16000 --seconds 3 --ragged
timeout 900 python tests/stage_sweep.py --n 100 --first-seed 1970000
