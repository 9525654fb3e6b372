bash tools/evidence_round.sh r05_d > gpurun_out/r05_d_evidence.log 2>&1
tail -20 gpurun_out/r05_d_evidence.log
