f() { python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', {k: round(v['ms'],1) for k,v in d['with_transfers'].items()})"; }
export WC_BENCH_SKIP=cheaptrick_config3,config2_16k_full_pipeline,config4_synthesis_only_share,config5_streams_share,dropin_single_utterance
f none
export WC_BENCH_SKIP=config2_16k_full_pipeline,config4_synthesis_only_share,config5_streams_share,dropin_single_utterance
f c3
export WC_BENCH_SKIP=cheaptrick_config3,config4_synthesis_only_share,config5_streams_share,dropin_single_utterance
f c2
export WC_BENCH_SKIP=cheaptrick_config3,config2_16k_full_pipeline,config5_streams_share,dropin_single_utterance
f c4
export WC_BENCH_SKIP=cheaptrick_config3,config2_16k_full_pipeline,config4_synthesis_only_share,dropin_single_utterance
f c5
export WC_BENCH_SKIP=cheaptrick_config3,config2_16k_full_pipeline,config4_synthesis_only_share,config5_streams_share
f dropin
