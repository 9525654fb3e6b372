bash tools/evidence_round.sh r05_c > gpurun_out/r05_c_evidence.log 2>&1
tail -12 gpurun_out/r05_c_evidence.log | cut -c1-400
