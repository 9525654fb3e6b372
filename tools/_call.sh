bash tools/evidence_round.sh r05_f > gpurun_out/r05_f_evidence.log 2>&1
tail -14 gpurun_out/r05_f_evidence.log
