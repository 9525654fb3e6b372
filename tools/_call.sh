bash tools/evidence_round.sh r05_e > gpurun_out/r05_e_evidence.log 2>&1
tail -20 gpurun_out/r05_e_evidence.log
