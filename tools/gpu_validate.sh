#!/bin/bash
# One gpurun call that reproduces the round's evidence.  Outputs under gpurun_out/validate/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_validate.sh'          GPU tests, smoke, driver-form bench, rocprof summary +
#                                                                   by-stage split, PMC traffic + MFMA utilisation
#   gpurun --timeout 2400 -- 'bash tools/gpu_validate.sh all'      + the other full-size configs and the acceptance sweep
#   bash tools/gpu_validate.sh tp N                                 on an N-GPU node: the tensor-parallel path, both
#                                                                   exchanges A/B, FAILS (rc != 0) on any silent fallback
O=gpurun_out/validate; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
if [ "$1" = "tp" ]; then
  N=${2:-8}; rc=0
  # (0) message-passing litmus of the exchange across the N devices (< 60 s each): READY/DONE form, then alternating halves;
  #     a failure COUNT, not a hang — run before anything that trusts the exchange (assumptions 1-3)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29541 tools/xgmi_litmus.py > $O/litmus_tp${N}.json 2> $O/litmus_tp${N}.err || rc=1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29542 tools/xgmi_litmus.py --alternate > $O/litmus_tp${N}_alt.json 2> $O/litmus_tp${N}_alt.err || rc=1
  # ... and of the FUSED exchange (the engine's default at world > 1), fence-free and fenced form
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29543 tools/xgmi_litmus.py --xchg --iters 200000 > $O/litmus_tp${N}_xchg.json 2> $O/litmus_tp${N}_xchg.err || rc=1
  tail -c 600 $O/litmus_tp${N}.json $O/litmus_tp${N}_alt.json $O/litmus_tp${N}_xchg.json
  # (1) one-shot exchange REQUIRED (no silent RCCL fallback), whole-forward hipGraphs REQUIRED, world size checked
  python bench.py --gpus $N --steps 20 --warmup 5 --allreduce oneshot --require-graph-form whole > $O/bench_tp${N}_oneshot.json 2> $O/bench_tp${N}_oneshot.err || rc=1
  # (2) the same run over RCCL, for the A/B
  python bench.py --gpus $N --steps 20 --warmup 5 --allreduce rccl > $O/bench_tp${N}_rccl.json 2> $O/bench_tp${N}_rccl.err || rc=1
  # (3) the one-shot exchange with alternating staging halves (no DONE handshake) — opt-in until it has a multi-GPU number
  TRIFORCE_AR_ALTERNATE=1 python bench.py --gpus $N --steps 20 --warmup 5 --allreduce oneshot --require-graph-form whole > $O/bench_tp${N}_oneshot_alt.json 2> $O/bench_tp${N}_oneshot_alt.err || rc=1
  # (4) replicated decisions (no record broadcasts) — opt-in; same tokens per step as (1) expected, and
  #     `decisions` in the line says how many stream-digest checks passed across the ranks
  TRIFORCE_TP_REPLICATED_DECISIONS=1 TRIFORCE_TP_REPLICA_CHECK_EVERY=4 python bench.py --gpus $N --steps 20 --warmup 5 --allreduce oneshot --require-graph-form whole > $O/bench_tp${N}_replicated.json 2> $O/bench_tp${N}_replicated.err || rc=1
  python - "$N" $O/bench_tp${N}_oneshot.json $O/bench_tp${N}_rccl.json $O/bench_tp${N}_oneshot_alt.json $O/bench_tp${N}_replicated.json <<'PY' || rc=1
import json, sys
n, bad = int(sys.argv[1]), []
for path, want in ((sys.argv[2], "one-shot"), (sys.argv[3], "rccl"), (sys.argv[4], "one-shot"), (sys.argv[5], "one-shot")):
    try:
        j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception as e:
        bad.append(f"{path}: no JSON line ({e})"); continue
    if j.get("failed"): bad.append(f"{path}: {j['failed']}")
    if j.get("config", {}).get("world_size_observed") != n: bad.append(f"{path}: RCCL saw world {j.get('config', {}).get('world_size_observed')} != {n}")
    if not str(j.get("decode_allreduce", "")).startswith(want): bad.append(f"{path}: decode_allreduce = {j.get('decode_allreduce')!r}, wanted {want}")
    if j.get("allreduce_error"): bad.append(f"{path}: allreduce_error {j['allreduce_error']}")
    if want == "one-shot" and j.get("graph_form") != "whole": bad.append(f"{path}: graph_form {j.get('graph_form')}")
    print(path, {k: j.get(k) for k in ("value", "ms_per_step", "tokens", "graph_form", "decode_allreduce", "allreduce_error", "acceptance_rate", "decisions")})
print(json.dumps({"tp_validate_ok": not bad, "problems": bad}))
sys.exit(1 if bad else 0)
PY
  echo "tp validate rc=$rc"; exit $rc
fi
rm -f gpurun_out/parity_notes.txt gpurun_out/parity_table.jsonl
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
cp gpurun_out/parity_notes.txt $O/parity_notes.txt 2>/dev/null; grep -c "" $O/parity_notes.txt
# the tolerance table: every recorded check -> measured max / mean |d|, bound, slack
cp gpurun_out/parity_table.jsonl $O/parity_table.jsonl 2>/dev/null; python tools/parity_table.py $O/parity_table.jsonl $O/parity_table.md --json $O/parity_table.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 800 $O/bench.json
# 200 timed steps: ms_per_step as mean +- stdev over 10 blocks of 20 (what a sub-1 % claim is read against)
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --random-steps 0 --roofline-every 0 > $O/bench_200.json 2> $O/bench_200.err; echo "bench200 rc=$?"; python -c "import json;j=json.loads([l for l in open('$O/bench_200.json') if l.startswith('{')][-1]);print(j['value'],j['ms_per_step'],j.get('ms_per_step_blocks'))"
# the draft step: 13-launch chain against the one-launch form (hipGraph replays), with the per-role timeline
python tools/draft_persist_bench.py --rows 1 3 7 --stamps --tag validate --out $O/draft_persist.jsonl > $O/draft_persist.log 2>&1; echo "draft persist rc=$?"; grep -c "" $O/draft_persist.jsonl
# temperature + top-p + softmax: one workgroup per row against the multi-workgroup launch
rm -f $O/topp_bench.jsonl; python tools/topp_bench.py --tag validate --out $O/topp_bench.jsonl > $O/topp_bench.log 2>&1; echo "topp bench rc=$?"; grep -c "" $O/topp_bench.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err; echo "rocprof rc=$?"
# target-verify / retrieval-verify launches of the split-KV kernel separated by duration cluster + priced (tracked copy -> profiles/)
T=$(ls -S $R/$O/prof/*/*kernel_trace.csv | head -1)
python $R/tools/attn_by_grid.py $T $R/$O/kernels_by_stage.json "rocprofv3 --kernel-trace of python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 (tools/gpu_validate.sh)" > $R/$O/kernels_by_stage.log 2>&1; echo "by-stage rc=$?"
python $R/tools/gap_analysis.py $T --steps 19 > $R/$O/gap_analysis_decode_steps.txt 2>&1; head -3 $R/$O/gap_analysis_decode_steps.txt
# HBM traffic of the roofline kernel (two separate --pmc passes, kernel-trace only) -> the figure bench.py quotes as roofline.traffic
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python $R/tools/pmc_attn.py > $R/$O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python $R/tools/pmc_attn.py > $R/$O/pmc_write.log 2>&1; echo "pmc write rc=$?"
# MFMA-pipe utilisation of the same kernel (own pass: SQ + GRBM counters only)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/$O/pmc_mfma -- python $R/tools/pmc_attn.py > $R/$O/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
# ... and of every kernel of one retrieval-verify decoder layer (GEMMs + retrieval attention), same two-pass recipe
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_layer_fetch -- python $R/tools/pmc_layer.py > $R/$O/pmc_layer_fetch.log 2>&1; echo "pmc layer fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_layer_write -- python $R/tools/pmc_layer.py > $R/$O/pmc_layer_write.log 2>&1; echo "pmc layer write rc=$?"
cd $R; python tools/pmc_reduce.py $O/pmc_fetch $O/pmc_write $O/pmc_attn_target_verify.json "tools/gpu_validate.sh"
python tools/pmc_layer_reduce.py $O/pmc_layer_fetch $O/pmc_layer_write gpurun_out/pmc_layer_meta.json $O/pmc_retrieval_verify_layer.json "tools/gpu_validate.sh"
python tools/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma_attn_target_verify.json "tools/gpu_validate.sh"
find $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_layer_fetch $O/pmc_layer_write -name "*kernel_trace.csv" -size +20M -delete
# the other full-size configs (BASELINE configs[2], and configs[3] at world size 1: offloading tier), only with "all"
if [ "$1" = "all" ]; then
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 --eager-comparator > $O/bench_eager_comparator.json 2> $O/bench_eager_comparator.err; echo "eager comparator rc=$?"
  python bench.py --target lwm-128K --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $O/bench_lwm.json 2> $O/bench_lwm.err; echo "lwm rc=$?"; tail -c 300 $O/bench_lwm.json
  python bench.py --prefill 130048 --budget 12288 --gamma 16 --on-chip 9 --steps 8 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench_offload.json 2> $O/bench_offload.err; echo "offload rc=$?"; tail -c 300 $O/bench_offload.json
  # (the 9-point acceptance sweep costs ~6 GPU-minutes: only with SWEEP=1; the tracked sweep is profiles/r04_acceptance_sweep.json)
  if [ "$SWEEP" = "1" ]; then python tools/acceptance_sweep.py $O/acceptance_sweep.json > $O/acceptance_sweep.log 2>&1; echo "sweep rc=$?"; tail -3 $O/acceptance_sweep.log; fi
  python tools/verify_bench.py final > $O/verify_bench.json 2> $O/verify_bench.err; echo "verify_bench rc=$?"; tail -c 400 $O/verify_bench.json
  # BASELINE configs[4] parameters at world size 1 (13B, gamma 16): the single-GPU line of the gamma = 16 path
  python bench.py --target llama-13B-128K --prefill 130048 --budget 12288 --gamma 16 --steps 10 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench_13b_cfg4.json 2> $O/bench_13b_cfg4.err; echo "13B rc=$?"; tail -c 300 $O/bench_13b_cfg4.json
  # BASELINE configs[3] parameters with every layer resident (the loop statistics of the gamma = 16 7B line of the prediction)
  python bench.py --prefill 130048 --budget 12288 --gamma 16 --steps 10 --warmup 2 --no-cpu-baseline --random-steps 0 > $O/bench_7b_cfg3_resident.json 2> $O/bench_7b_cfg3_resident.err; echo "7B cfg3 resident rc=$?"; tail -c 300 $O/bench_7b_cfg3_resident.json
  # per-rank stage latencies at W = 1 / 2 / 4 / 8 for configs[1] / [3] / [4] (rank 0's shard on this GPU, exchanges in their
  # shipped form against a one-rank group) -> the predicted scaling table the first multi-GPU run is judged against
  rm -f $O/tp_shard_by_world.jsonl
  for W in 1 2 4 8; do
    X="--local-exchange --gemm-exchange"; [ $W = 1 ] && X=""
    python tools/tp_shard_bench.py llama-7B-128K $W --gamma 6 --prefill 124928 --budget 4096 $X 2>>$O/tp_shard.err | grep '^{' >> $O/tp_shard_by_world.jsonl
    python tools/tp_shard_bench.py llama-7B-128K $W --gamma 16 --prefill 130048 --budget 12288 $X 2>>$O/tp_shard.err | grep '^{' >> $O/tp_shard_by_world.jsonl
    python tools/tp_shard_bench.py llama-13B-128K $W --gamma 16 --prefill 130048 --budget 12288 $X 2>>$O/tp_shard.err | grep '^{' >> $O/tp_shard_by_world.jsonl
  done
  # the TP engine's own loop at world 1 on the headline workload: what its host path costs on top of the graph engine's
  python bench.py --engine tp --steps 20 --warmup 5 --no-cpu-baseline --random-steps 0 > $O/bench_tp_world1.json 2> $O/bench_tp_world1.err; echo "tp world1 rc=$?"
  # loop statistics (tokens per step, inner iterations, host overhead) and the W = 1 row from THIS run's single-GPU lines
  python tools/predict_scaling.py --shards $O/tp_shard_by_world.jsonl --out $O/predicted_scaling.json \
      --bench "configs[1]=$O/bench.json" "configs[3]=$O/bench_7b_cfg3_resident.json" "configs[4]=$O/bench_13b_cfg4.json" --tp-host "configs[1]=$O/bench_tp_world1.json" > $O/predicted_scaling.md; echo "predict rc=$?"; head -8 $O/predicted_scaling.md
  (cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_tp8 -- python $R/tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange > $R/$O/prof_tp8.log 2>&1)
  T8=$(ls -S $O/prof_tp8/*/*kernel_trace.csv | head -1)
  python tools/kernel_timeline.py $T8 $O/tp8_7b_kernel_timeline.json "rocprofv3 --kernel-trace of tools/tp_shard_bench.py llama-7B-128K 8 --gamma 6 --prefill 124928 --budget 4096 --local-exchange --gemm-exchange (rank 0 shard of an 8-way 7B engine on one MI355X), final build of the round (tools/gpu_validate.sh all)" > $O/tp8_7b_kernel_timeline.txt 2>&1
  find $O/prof_tp8 -name "*kernel_trace.csv" -size +20M -delete
fi
