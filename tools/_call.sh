bash tools/evidence_round.sh r05_b > gpurun_out/r05_b_evidence.log 2>&1
tail -12 gpurun_out/r05_b_evidence.log | cut -c1-400
