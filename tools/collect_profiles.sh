#!/bin/bash
# Copies what tools/evidence_round.sh TAG left under gpurun_out/ into profiles/ under the names profiles/README.md lists:
#   bash tools/collect_profiles.sh r04_a
set -u
T=$1
G=gpurun_out
P=profiles
cp $G/${T}_bench.json $P/${T}_bench_as_run.json
cp $G/$T/stats_bench.json $P/${T}_bench_under_rocprof.json
cp $G/$T/stats/p_kernel_stats.csv $P/${T}_kernel_stats.csv
cp $G/$T/stats/p_domain_stats.csv $P/${T}_domain_stats.csv
cp $G/$T/stats_config3/p_kernel_stats.csv $P/${T}_config3_kernel_stats.csv
cp $G/$T/stats_config3.txt $P/${T}_config3_microbench.txt
cp $G/$T/sq_counters.txt $P/${T}_sq_counters.txt
cp $G/$T/sq_counters_config3.txt $P/${T}_config3_sq_counters.txt
cp $G/$T/isa_mix.txt $P/${T}_isa_mix.txt
cp $G/$T/kernel_resources.txt $P/${T}_kernel_resources.txt
cp $G/$T/pmc_traffic.json $P/${T}_pmc_traffic.json
cp $G/$T/pmc_traffic.json $P/pmc_traffic.json
cp $G/$T/pmc_traffic_config3.json $P/${T}_pmc_traffic_config3.json
grep -E 'passed|failed|FAILED' $G/${T}_gputest.log | tail -5 > $P/${T}_gputest.log
cp $G/${T}_mb64.txt $P/${T}_stage_by_stage_64utt.txt
cp $G/${T}_mb16.txt $P/${T}_stage_by_stage_16utt.txt
cp $G/${T}_mb1.txt $P/${T}_stage_by_stage_1utt.txt
cp $G/${T}_latency.txt $P/${T}_latency_probe.txt
cp $G/${T}_issue_rates.txt $P/${T}_issue_rates.txt
cp $G/${T}_issue_rates.json $P/${T}_issue_rates.json
cp $G/${T}_issue_rates.json $P/issue_rates.json
grep -v "^run\|amdgpu.ids" $G/${T}_host_frontend_timing.txt | tail -40 > $P/${T}_host_frontend_timing.txt
ls -la $P | grep $T
