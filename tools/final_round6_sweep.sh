#!/bin/bash
# The sweep on the LAST build of round 6 (scan decimation, refinement in frame groups, raw candidates by slice descriptors, ties confined
# to their utterances): the shapes of tools/final_round5_sweep.sh with seeds of their own; every summary row now ends with the number of
# utterances the tie flag sent through the FIR twin.  The last three rows are checked by the REAL reference (a fresh process per signal).
#   bash tools/final_round6_sweep.sh > gpurun_out/sweep8.txt 2>&1
S="timeout 1700 python tests/parity_sweep.py"
$S --n 300 --first-seed 2900000 --fs 48000 --seconds 10 --ragged
$S --n 200 --first-seed 2910000 --fs 16000 --seconds 8 --ragged --floor 40
$S --n 100 --first-seed 2920000 --fs 44100 --seconds 5 --ragged --frame-period 1
$S --n 100 --first-seed 2930000 --fs 24000 --seconds 4 --ragged --frame-period 1
$S --n 100 --first-seed 2940000 --zoo --fs 16000 --seconds 3
$S --n 60 --first-seed 2950000 --zoo --dither 1e-3 --fs 48000 --seconds 2
$S --n 40 --first-seed 2960000 --fs 32000 --seconds 3 --ragged
$S --n 60 --first-seed 2980000 --zoo2 --dither 1e-3 --fs 48000 --seconds 2
$S --n 60 --first-seed 2990000 --zoo2 --dither 1e-4 --fs 16000 --seconds 3 --ragged
timeout 900 python tests/stage_sweep.py --n 100 --first-seed 2970000
$S --n 60 --first-seed 3000000 --zoo --fs 48000 --seconds 2 --checker ref --nan-tolerant
$S --n 60 --first-seed 3010000 --zoo2 --fs 24000 --seconds 3 --checker ref --nan-tolerant
$S --n 40 --first-seed 3020000 --fs 96000 --seconds 3 --checker ref --nan-tolerant
