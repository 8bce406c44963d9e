#!/bin/bash
# Copy the summaries of a `bash tools/gpu_validate.sh all` run (gpurun_out/validate/) into profiles/ under the round's names.
#   bash tools/collect_profiles.sh r05
R=${1:?round prefix, e.g. r05}; V=gpurun_out/validate; P=profiles
cp $V/bench.json $P/${R}_bench_default_n1.json
cp $V/bench_prof.json $P/${R}_bench_default_under_rocprof.json
S=$(ls -S $V/prof/*/*kernel_stats.csv | head -1); cp $S $P/${R}_kernel_stats_bench_default.csv
cp $V/kernels_by_stage.json $P/${R}_kernels_by_stage_bench_default.json
cp $V/gap_analysis_decode_steps.txt $P/${R}_gap_analysis_decode_steps.txt
cp $V/pmc_attn_target_verify.json $P/${R}_pmc_attn_target_verify.json
cp $V/pmc_mfma_attn_target_verify.json $P/${R}_pmc_mfma_attn_target_verify.json
cp $V/pmc_retrieval_verify_layer.json $P/${R}_pmc_retrieval_verify_layer.json
cp $V/parity_notes.txt $P/${R}_parity_notes.txt
[ -s $V/parity_table.md ] && cp $V/parity_table.md $P/${R}_parity_table.md && cp $V/parity_table.json $P/${R}_parity_table.json
[ -s $V/bench_200.json ] && cp $V/bench_200.json $P/${R}_bench_default_200_steps.json
[ -s $V/draft_persist.jsonl ] && cp $V/draft_persist.jsonl $P/${R}_draft_persist_chain_vs_one_launch.jsonl
[ -s $V/topp_bench.jsonl ] && cp $V/topp_bench.jsonl $P/${R}_topp_one_workgroup_vs_multi.jsonl
[ -s $V/bench_eager_comparator.json ] && cp $V/bench_eager_comparator.json $P/${R}_bench_eager_comparator_n1.json
[ -s $V/acceptance_sweep.json ] && cp $V/acceptance_sweep.json $P/${R}_acceptance_sweep.json
# provenance: the commit these figures were measured on (the GPU box has no .git: stamped here, at collection time)
H=$(git rev-parse --short HEAD 2>/dev/null)
for f in $P/${R}_pmc_attn_target_verify.json $P/${R}_pmc_retrieval_verify_layer.json $P/${R}_acceptance_sweep.json; do
  [ -s $f ] && python - "$f" "$H" <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); j["measured_on"] = sys.argv[2]; json.dump(j, open(sys.argv[1], "w"), indent=1)
PY
done
tail -12 $V/pytest.log | grep -a "passed\|failed" > $P/${R}_gpu_pytest_summary.txt; tail -1 $V/smoke.log >> $P/${R}_gpu_pytest_summary.txt
for f in bench_lwm:bench_lwm_128k_full_n1 bench_offload:bench_offload_cfg3_world1 bench_13b_cfg4:bench_13b_cfg4_world1 bench_7b_cfg3_resident:bench_7b_cfg3_resident_world1 bench_tp_world1:bench_tp_engine_world1 verify_bench:verify_bench_final; do
  [ -s $V/${f%%:*}.json ] && cp $V/${f%%:*}.json $P/${R}_${f#*:}.json
done
[ -s $V/tp_shard_by_world.jsonl ] && cp $V/tp_shard_by_world.jsonl $P/${R}_tp_shard_by_world.jsonl
[ -s $V/predicted_scaling.json ] && cp $V/predicted_scaling.json $P/${R}_predicted_scaling.json
[ -s $V/tp8_7b_kernel_timeline.json ] && cp $V/tp8_7b_kernel_timeline.json $P/${R}_tp8_7b_kernel_timeline_final.json
ls -la $P/${R}_* | wc -l
