#!/bin/bash
# A second sweep on the LAST build of round 6, seeds and shapes no earlier sweep has seen (22.05 / 11.025 / 8 kHz, 2 ms hop, floors 50 / 90,
# longer utterances); the last three rows against the REAL reference (a fresh process per signal):
#   bash tools/final_round6_sweep2.sh > gpurun_out/sweep12.txt 2>&1
S="timeout 1700 python tests/parity_sweep.py"
$S --n 200 --first-seed 4900000 --fs 48000 --seconds 12 --ragged --floor 50
$S --n 150 --first-seed 4910000 --fs 22050 --seconds 6 --ragged
$S --n 100 --first-seed 4920000 --fs 16000 --seconds 10 --frame-period 2 --floor 90
$S --n 100 --first-seed 4930000 --fs 24000 --seconds 6 --ragged --frame-period 2
$S --n 60 --first-seed 4940000 --zoo --dither 1e-3 --fs 24000 --seconds 3
$S --n 60 --first-seed 4950000 --zoo2 --dither 1e-3 --fs 22050 --seconds 3 --ragged
$S --n 60 --first-seed 4960000 --fs 44100 --seconds 4 --ragged --floor 40
timeout 900 python tests/stage_sweep.py --n 100 --first-seed 4970000
$S --n 40 --first-seed 5000000 --fs 11025 --seconds 4 --checker ref --nan-tolerant
$S --n 40 --first-seed 5010000 --fs 8000 --seconds 4 --checker ref --nan-tolerant
$S --n 60 --first-seed 5020000 --zoo2 --dither 1e-3 --fs 48000 --seconds 2 --checker ref --nan-tolerant
