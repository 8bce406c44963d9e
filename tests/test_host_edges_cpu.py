"""Edge cases of the host logic on CPU (oracle-backed ops): EOS inside the accepted chain, capacity limits the
reference enforces by crashing, config validation, the uniform stream, synthetic data."""
import os
import pytest
import torch

from oracle import ref_model as M
from tests import helpers as Hh


@pytest.mark.parametrize("eos_pick", [3, 6, 9, 14])
def test_eos_paths_match_oracle(cpu_ops, eos_pick):
    """Whatever token is declared EOS, product and oracle take the same branch (decoding.py:108-110,120-121)."""
    from triforce_amd.utils.decoding import TriForce
    from triforce_amd.utils.sampling import UniformSource
    g = Hh.load_golden("small_gamma6")
    us = Hh.fixed_uniforms(seed=7)
    oeng, tsd, dsd = Hh.build_oracle(g, temperature=0.6, top_p=0.9)
    prompt = Hh.prompt_of(g)
    base = M.triforce(oeng, prompt, g["gamma"], 20, 0.6, 0.9, rng=M.InjectedRng(us), eos_token_id=-1)
    eos = base["tokens"][eos_pick]                      # a token that really occurs in the stream
    oeng, _, _ = Hh.build_oracle(g, temperature=0.6, top_p=0.9)       # fresh engine: a 2nd prompt carries state over
    want = M.triforce(oeng, prompt, g["gamma"], 20, 0.6, 0.9, rng=M.InjectedRng(us), eos_token_id=eos)
    tok = Hh.FakeTokenizer()
    tok.eos_token_id = eos
    ge = Hh.build_product(g, "cpu", tsd, dsd, temperature=0.6, top_p=0.9)
    got = TriForce(tok, ge, prompt, gamma=g["gamma"], max_len=20, top_k=-1, top_p=0.9, temperature=0.6,
                   rng=UniformSource("cpu", values=us), return_details=True)
    assert got["tokens"] == want["tokens"] and got["counts"] == want["counts"]
    assert got["drafted"] == want["drafted"] and got["accepted"] == want["accepted"]


def test_retrieval_tail_overflow_raises_like_the_reference(cpu_ops):
    """gen_len + gamma + 1 > budget makes the reference crash in cache.py:181; here it is an
    explicit IndexError instead of a shape-mismatch RuntimeError."""
    from triforce_amd.models.cache import FlashSimpleCache, RetrievalCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    from oracle import specs
    cfg = specs.tiny_target_config(vocab_size=128, layers=1, hidden=128, heads=1, max_pos=512)
    m = LlamaForCausalLM.from_state_dict(LlamaConfig.from_dict(cfg), specs.random_state_dict(cfg, 1), "cpu")
    kv = FlashSimpleCache(m, 64 + 40)
    rc = RetrievalCache(m, max_budget=16, prefill=64, gamma=2, chunk_size=8)
    kv.seq_len = 64 + 17                                 # 17 generated tokens > budget 16
    with pytest.raises(IndexError):
        rc.update_graph_cache(kv)
    kv.seq_len = 64 + 16
    rc.update_graph_cache(kv)                            # exactly full is fine
    with pytest.raises(IndexError):
        kv.seq_len = 64 + 40
        kv.append_slot(0, 1)                             # full-cache overflow (on_chip.py:78 slack exhausted)
    with pytest.raises(AssertionError):
        RetrievalCache(m, max_budget=16, prefill=60, gamma=2, chunk_size=8)     # prefill % chunk != 0 (cache.py:126)


def test_config_validation():
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models import zoo
    c = zoo.config("llama-7B-128K")
    assert (c.hidden_size, c.num_hidden_layers, c.head_dim, c.rope_scaling["factor"]) == (4096, 32, 128, 32.0)
    assert zoo.config("llama-13B-128K").intermediate_size == 13824 and zoo.config("lwm-128K").rope_theta == 1e7
    with pytest.raises(ValueError):
        LlamaConfig(num_attention_heads=8, num_key_value_heads=2)        # GQA unsupported
    with pytest.raises(ValueError):
        LlamaConfig(rope_scaling={"type": "linear", "factor": 2.0})
    with pytest.raises(ValueError):
        LlamaConfig(rope_scaling={"type": "yarn", "factor": 0.5, "original_max_position_embeddings": 4096})
    with pytest.raises(NotImplementedError):
        zoo.config("gpt-17")


def test_uniform_source_matches_injected_rng_across_wraparound():
    from triforce_amd.utils.sampling import UniformSource
    vals = [i / 97.0 for i in range(97)]
    src = UniformSource("cpu", values=vals, block=128)
    ref = M.InjectedRng(vals)
    for k in [3, 1, 7, 60, 64, 2, 33, 64, 64, 5]:
        got = src.take(k).tolist()
        want = [ref._next() for _ in range(k)]
        assert got == pytest.approx(want)
        src.advance(k)


def test_uniform_source_device_cursor_mirrors_the_host_position():
    """Round 5: kernels captured inside a hipGraph read their uniforms as buf[cursor + k] (ops.*_cur) and advance the device
    cursor themselves; the host mirrors it.  Pinned here (CPU tensors stand in for the device): the buffer keeps its ADDRESS
    across refills and wrap-arounds (captured graphs hold it), cursor_tensor() brings a stale device copy up to date and makes
    room first, advanced_on_device() checks the kernel's report against the mirror, and the numbers a cursor reader would see
    are the numbers take() hands out."""
    from triforce_amd.utils.sampling import UniformSource
    src = UniformSource("cpu", seed=7, block=64)
    ref = UniformSource("cpu", seed=7, block=64)
    addr, caddr = src.buf.data_ptr(), src.cursor.data_ptr()
    for k in [3, 5, 2, 40, 9, 3, 30, 3, 3, 50, 7]:
        want = ref.take(k).clone()
        ref.advance(k)
        cur = src.cursor_tensor(k)                                # (may refill in place and reset the position)
        assert cur.data_ptr() == caddr and src.buf.data_ptr() == addr and int(cur) == src.pos and src.device_cursor
        got = src.buf[int(cur):int(cur) + k]
        assert torch.equal(got, want)
        at = int(cur)
        src.cursor += k                                           # what the *_cur kernel that closes the decision does
        src.advanced_on_device(k, at)
        assert int(src.cursor) == src.pos
    src.take(2)
    src.advance(2)                                                # an eager kernel consumed two numbers through a pointer
    assert not src.device_cursor and int(src.cursor) != src.pos
    assert int(src.cursor_tensor(3)) == src.pos and src.device_cursor
    with pytest.raises(RuntimeError, match="out of step"):
        src.advanced_on_device(3, at=src.pos + 1)
    fixed = UniformSource("cpu", values=[i / 13.0 for i in range(13)], block=32)
    seen = []
    for k in [5, 9, 11, 7, 12]:
        c = int(fixed.cursor_tensor(k))
        seen += fixed.buf[c:c + k].tolist()
        fixed.cursor += k
        fixed.advanced_on_device(k, c)
    assert seen == pytest.approx([(i % 13) / 13.0 for i in range(len(seen))])


def test_synthetic_dataset_and_null_tokenizer():
    from triforce_amd.data.dataset import NullTokenizer, get_dataset, load_tokenizer
    a = get_dataset("synthetic", datalen=100, vocab_size=500, num_prompts=2, seed=3)
    b = get_dataset("synthetic", datalen=100, vocab_size=500, num_prompts=2, seed=3)
    assert len(a) == 2 and a[0].shape == (1, 100) and torch.equal(a[1], b[1]) and int(a[0].min()) >= 3
    tok = load_tokenizer("none", 500)
    assert isinstance(tok, NullTokenizer) and tok.decode([1, 2]) == ""
    with pytest.raises(Exception):
        get_dataset("no-such-dataset")


def test_middle_spec_standalone_signature(cpu_ops):
    """Middle_Spec(next_token, graph_engine, gamma, verbose, tokenizer) is callable exactly like the reference's."""
    from triforce_amd.utils.decoding import Middle_Spec
    g = Hh.load_golden("small_gamma6")
    ge = Hh.build_product(g, "cpu")
    prompt = Hh.prompt_of(g)
    ge.engine.kv_cache.reset()
    ge.inference(prompt[:, :-1])
    ge.inference(prompt[:, -1:])
    ge.graph_draft_prefill(prompt)
    ids, rows, acc = Middle_Spec(torch.tensor([[5]]), ge, g["gamma"], False, Hh.FakeTokenizer())
    assert ids[0] == 5 and g["gamma"] <= len(ids) - 1 <= g["gamma"] + 1
    assert rows.shape == (len(ids) - 1, g["tcfg"]["vocab_size"]) and 0.0 <= acc <= 1.0
    assert torch.allclose(rows.sum(-1), torch.ones(len(ids) - 1), atol=1e-5)


def test_cli_flag_tables_match_the_reference_scripts():
    """Drop-in boundary of the entry scripts: every flag of the reference's test/*.py exists here with the same type
    and default (tests/golden/cli_flags.json, read from the reference's add_argument calls by oracle/gen_golden.py).
    Deliberate differences: --dataset defaults to `synthetic` (the reference's datasets are absent offline), and
    offline-only flags (--weights, --draft-weights, --tokenizer, --greedy, --no_graphs) and --rebuild_every (SURVEY 8f
    row 4, off by default) are additions."""
    import json
    import os
    from triforce_amd.utils import cli
    here = os.path.dirname(os.path.abspath(__file__))
    ref = json.load(open(os.path.join(here, "golden", "cli_flags.json")))
    names = {int: "int", float: "float", str: "str"}
    for script, flags in ref.items():
        ours = {name: (typ if typ == "flag" else names[typ], default) for name, typ, default, _ in cli.SCRIPTS[script]}
        for flag, (typ, default) in flags.items():
            assert flag in ours, f"{script}: reference flag {flag} missing"
            assert ours[flag][0] == typ, f"{script} {flag}: type {ours[flag][0]} != {typ}"
            if flag == "--dataset":
                assert ours[flag][1] == "synthetic"
            elif typ != "flag":
                assert ours[flag][1] == default, f"{script} {flag}: default {ours[flag][1]!r} != {default!r}"
        extra = set(ours) - set(flags)
        assert extra <= {"--weights", "--draft-weights", "--tokenizer", "--greedy", "--no_graphs", "--file",
                         "--rebuild_every"}, (script, extra)
        args = cli.parse(script, [])                      # every table parses with its defaults
        assert args.gen_len == 256 and args.temp == 0.6 and args.top_p == 0.9
    assert cli.parse("on_chip", ["--greedy"]).top_p == 1e-9 and cli.parse("on_chip", ["--greedy"]).temp == 1.0
    assert cli.parse("offloading_TP", ["--gamma", "16"]).gamma == "16"      # the reference parses --gamma as str here


@pytest.mark.parametrize("n", [65, 127, 128, 129, 1000, 1024, 1025, 2047, 2048, 3001])
@pytest.mark.parametrize("on_device", [False, True])
def test_chunked_prefill_returns_the_reference_last_chunk(n, on_device):
    """The device feeds the prompt PREFILL_CHUNK rows per forward, the reference 128 (graph_infer.py:30-37): whatever the
    chunking, every token goes through exactly once, in order, and the returned logits are the rows of the REFERENCE's
    last 128-token chunk."""
    from triforce_amd.utils import graph_infer as gi

    class Ids:                                        # shape / slicing / is_cuda are all chunked_prefill touches
        def __init__(self, lo, hi):
            self.lo, self.hi, self.is_cuda, self.shape = lo, hi, on_device, (1, hi - lo)

        def __getitem__(self, key):
            sl = key[1]
            lo = self.lo + (sl.start or 0)
            return Ids(lo, min(self.hi, self.lo + sl.stop) if sl.stop is not None else self.hi)

    seen = []

    def forward(chunk):
        seen.append((chunk.lo, chunk.hi))
        return torch.arange(chunk.lo, chunk.hi, dtype=torch.float32).reshape(1, -1, 1)

    out = gi.chunked_prefill(forward, Ids(0, n))
    step = gi.PREFILL_CHUNK if on_device else 128
    assert seen == [(i, min(n, i + step)) for i in range(0, n, step)]
    ref_last = 128 * ((n - 1) // 128)                  # first row of the reference's last chunk
    assert out.reshape(-1).tolist() == list(range(ref_last, n))
    # a forward that takes last_rows is asked for exactly the returned rows of the final chunk and ONE row of the others
    seen.clear()
    asked = []

    def trimming(chunk, last_rows=None):
        seen.append((chunk.lo, chunk.hi))
        asked.append(last_rows)
        return torch.arange(chunk.hi - last_rows, chunk.hi, dtype=torch.float32).reshape(1, -1, 1)

    out2 = gi.chunked_prefill(trimming, Ids(0, n))
    assert seen == [(i, min(n, i + step)) for i in range(0, n, step)]
    assert asked == [1] * (len(seen) - 1) + [n - ref_last]
    assert out2.reshape(-1).tolist() == list(range(ref_last, n))


def test_bench_helpers_on_cpu():
    """bench.py's host-side arithmetic: the roofline helper divides algorithmic bytes by the sampled launch durations,
    the KV slack is sized from the requested steps (timed + warm-up + the secondary random-weight run + the AR probe),
    and the weight resolution falls back to aligned synthetic weights when no checkpoint is on disk."""
    import bench

    class Ev:                                                          # stands in for a pair of HIP events
        def __init__(self, ms):
            self.ms = ms

        def elapsed_time(self, other):
            return other.ms

    timer = [(Ev(0), Ev(0.350), 124936, 32, 128), (Ev(0), Ev(0.022), 4103, 32, 128), (Ev(0), Ev(0.354), 124944, 32, 128)]
    roof = bench.attn_roofline(timer, retrieval_rows=4103, H=32, D=128)
    alg = (2 * 124936 * 32 * 128 * 2 + 2 * 124944 * 32 * 128 * 2) / 2
    assert roof["launches"] == 2 and roof["algorithmic_bytes_per_launch"] == int(alg)
    assert abs(roof["achieved"] - alg / 0.352e-3 / 1e9) < 0.2 and abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-3
    assert bench.attn_roofline(timer[1:2], retrieval_rows=4103, H=32, D=128) is None      # no full-KV launch sampled
    assert bench.attn_roofline(None, retrieval_rows=4103, H=32, D=128) is None

    a = bench.parse(["--steps", "200"])
    assert a.random_steps == 12
    assert a.gen_cap == (200 + a.warmup + 12 + 8) * (a.gamma + 2) + a.ar_steps + 64 and a.gen_cap <= a.budget
    assert bench.parse(["--steps", "5000"]).gen_cap == 4096            # never more than the retrieval budget
    assert bench.parse(["--steps", "20", "--random-steps", "0"]).random_steps == 0

    kind, tspec, dspec, label = bench.resolve_weights(bench.parse([]))   # no checkpoints in this container
    assert kind == "aligned" and tspec == dspec == "aligned:0.7:0.9"
    kind, tspec, dspec, _ = bench.resolve_weights(bench.parse(["--weights", "random:3"]))
    assert (kind, tspec, dspec) == ("random", "random:4", "random:5")
    kind, tspec, _, _ = bench.resolve_weights(bench.parse(["--weights", "aligned:0.5:0.95:7"]))
    # round 6: ms per step as mean +- stdev over 10 blocks of the timed steps (host stamps behind every step)
    stamps = [0.0]
    for i in range(200):
        stamps.append(stamps[-1] + (0.027 if i < 100 else 0.028))
    bs = bench.block_stats(stamps)
    assert bs["blocks"] == 10 and bs["steps_per_block"] == 20 and abs(bs["mean"] - 27.5) < 1e-6
    assert abs(bs["min"] - 27.0) < 1e-6 and abs(bs["max"] - 28.0) < 1e-6 and 0.52 < bs["stdev"] < 0.53
    assert bench.block_stats(stamps[:20]) is None                                   # fewer than 2 steps per block
    # ... and three other points of the acceptance dial from the newest committed sweep, with their provenance
    op = bench.operating_points("cfg")
    assert op is not None and op["source"].startswith("profiles/") and 1 <= len(op["points"]) <= 4
    assert all("tokens_per_s" in p and "requested_draft_acc" in p for p in op["points"])
    assert kind == "aligned" and tspec == "aligned:0.5:0.95:7"


def test_bench_self_launches_for_more_than_one_gpu():
    """``python bench.py --gpus 2`` without a launcher (the driver's command form) re-executes itself under
    torch.distributed.run with one rank per device, builds the process group and the sharded engine on every rank
    (gloo + CPU here, --dry-run stops before any kernel) and rank 0 alone prints the one JSON line, carrying the
    world size the process group observed.  Mirrors the reference's launch line (README.md:62,
    test/offloading_TP.py:23: torchrun --nproc_per_node=2)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--target", "tiny", "--prefill",
                          "512", "--budget", "64", "--gamma", "4", "--dry-run", "--weights", "random:3"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["dry_run"] and j["n_gpus"] == 2 and j["world_size_observed"] == 2 and j["backend"] == "gloo"
    assert [s[2] for s in j["shards"]] == [0, 1] and all(s[0] == 1 and s[1] == 384 for s in j["shards"])   # 2 heads, I=768
    assert j["config"]["weights"].startswith("random-init")
    # a launcher whose world size disagrees with --gpus is refused with a message, not an assertion trace
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                         text=True, timeout=120, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), cwd=root)
    assert bad.returncode != 0 and "WORLD_SIZE=4" in (bad.stderr + bad.stdout)


def test_multi_rank_report_schema_and_prediction_lookup(tmp_path):
    """The part of the N > 1 bench line the scaling prediction is written in (bench_tp.multi_rank_report; reference
    test/offloading_TP.py:104-119 prints its latencies from every configuration): schema pinned here on fake numbers, the
    prediction looked up from a tools/predict_scaling.py file by configuration label and world size."""
    import json
    import subprocess
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench_tp
    shards = tmp_path / "shards.jsonl"
    with open(shards, "w") as f:
        for W, (tv, rv, dr) in {1: (13400.0, 3280.0, 115.0), 2: (7600.0, 2300.0, 126.0), 8: (2760.0, 1440.0, 132.0)}.items():
            f.write(json.dumps({"target": "llama-7B-128K", "emulated_world": W, "heads_per_rank": 32 // W, "layers": 32,
                                "prefill": 124928, "budget": 4096, "gamma": 6, "draft_step_us": dr, "retrieval_verify_us": rv,
                                "target_verify_us": tv, "decode_layer": "fused (6 launches)", "exchange": "x"}) + "\n")
    bench_line = tmp_path / "bench.json"
    bench_line.write_text(json.dumps({"n_gpus": 1, "value": 216.0, "tokens_per_step": 5.85, "inner_iterations_per_step": 3.9,
                                      "step_overhead_us": 300.0, "stage_latency_us": {"draft_step_us": 110.0, "retrieval_verify_us": 3250.0,
                                                                                     "target_verify_us": 13300.0}}) + "\n")
    out = tmp_path / "pred.json"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "predict_scaling.py"), "--shards", str(shards), "--out", str(out),
                        "--bench", f"configs[1]={bench_line}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    pj = json.load(open(out))["configs"]["configs[1]"]
    assert pj["measured_per_rank_us"][0]["decode_layer"].startswith("single-GPU graph engine")     # W = 1 row = the product
    assert pj["loop"]["host_overhead_us"] == 300.0 and "bench.json" in pj["loop"]["source"]
    pred = bench_tp.predicted_for("BASELINE configs[1] shapes sharded TP=8", 8, path=str(out))
    assert pred["world"] == 8 and pred["per_rank_us"]["retrieval_verify_us"] == 1440.0
    assert pred["low"]["tokens_per_s"] > pred["high"]["tokens_per_s"] > 0
    assert bench_tp.predicted_for("custom (not a BASELINE.json config)", 8, path=str(out)) is None
    assert bench_tp.predicted_for("BASELINE configs[1] shapes sharded TP=4", 4, path=str(out)) is None        # no W = 4 shard line
    per_rank = [{"rank": r, "stages": {"draft_step_us": 130.0 + r, "retrieval_verify_us": 1500.0 + 10 * r, "target_verify_us": 2900.0 - r},
                 "exchange": {"gemm_with_exchange_us": 9.5, "gemm_alone_us": 5.0, "per_exchange_us": 4.5, "shape": "s"}} for r in range(8)]
    rep = bench_tp.multi_rank_report(per_rank, None, types.SimpleNamespace(num_hidden_layers=32), 8,
                                     {"inner_iterations_per_step": 3.9, "ms_per_step": 11.0}, "inside the GEMMs", pred)
    assert set(rep) == {"stage_latency_us_per_rank", "stage_latency_us_slowest_rank", "exchange", "measured_step_terms_us", "predicted",
                        "measured_minus_predicted_us"}
    assert rep["stage_latency_us_slowest_rank"] == {"draft_step_us": 137.0, "retrieval_verify_us": 1570.0, "target_verify_us": 2900.0}
    assert rep["exchange"]["per_forward"] == 64 and rep["exchange"]["per_step"] == round(64 * 4.9, 1)
    assert rep["measured_minus_predicted_us"]["retrieval_verify_us"] == 130.0
    t = rep["measured_step_terms_us"]
    assert abs(t["target_verify"] + t["retrieval_verify"] + t["draft"] + t["host_and_broadcasts"] - 11000.0) < 1.0
    # a rank without stage numbers (offloading tier: stages are not measured) leaves the slowest-rank fields empty
    rep2 = bench_tp.multi_rank_report([{"rank": 0, "stages": None, "exchange": None}], None, types.SimpleNamespace(num_hidden_layers=2),
                                      1, {"inner_iterations_per_step": 2.0, "ms_per_step": 1.0}, "none (one rank)")
    assert rep2["stage_latency_us_slowest_rank"] is None and "predicted" not in rep2


def test_tree_walk_limits_and_atomic_grow_map_cache(tmp_path):
    """The device-side Sequoia accept walk reads one uniform per examined child from a fixed window and records a
    bounded path: SpecTree refuses grow maps that could exceed either (_worst_walk); a grow map built on demand is
    published with an atomic rename, so concurrent torchrun ranks never read a half-written file."""
    import os
    from triforce_amd.utils import tree as T
    from triforce_amd.utils.SpecTree_TP import _worst_walk
    assert _worst_walk([[]]) == (0, 0)
    #        0 -> 1,2,3 ; 1 -> 4,5 ; 4 -> 6 ; 2 -> 7
    succ = [[1, 2, 3], [4, 5], [7], [], [6], [], [], []]
    assert _worst_walk(succ) == (3 + 2 + 1, 3)
    gm = T.load_grow_map(512)
    c, d = _worst_walk(gm["Successors"])
    assert c < 255 and d < 59                                   # the shipped 512-node tree fits with room to spare
    small = T.load_grow_map(24, cache_dir=str(tmp_path))
    files = os.listdir(tmp_path)
    assert files == ["24.json"], files                          # no temp file left behind
    again = T.load_grow_map(24, cache_dir=str(tmp_path))
    assert again["branches"] == small["branches"] and again["size"] == 24


def test_topp_candidate_cut_always_contains_the_crossing():
    """csrc/sampling.hip restricts the top-p histogram to entries >= cut = (1 - top_p) / (2 V), rounded down to a round-1
    bin edge of the fp32 pattern, and takes Z from ALL entries.  The kernel relies on the crossing of tau = floor(top_p * Z)
    lying among those candidates.  Restated here in numpy with the kernel's 2^-40 fixed-point masses: random, peaked and
    adversarial rows (as much mass as possible parked just below the cut), V up to 32768, top_p up to 0.999."""
    import numpy as np

    def fix(e):                                              # topp_fix: floor(e * 2^40) from the bit pattern
        return np.floor(e.astype(np.float64) * 2.0 ** 40).astype(np.uint64)

    rng = np.random.default_rng(0)
    for V in (1000, 32000, 32768):
        for top_p in (0.5, 0.9, 0.95, 0.99, 0.999):
            cut = np.float32((np.float32(1.0) - np.float32(top_p)) / np.float32(2.0 * V))
            cutpat = cut.view(np.uint32) & np.uint32(~((1 << 20) - 1) & 0xFFFFFFFF)
            cut_edge = cutpat.view(np.float32)
            rows = [np.exp(rng.normal(0, s, V).astype(np.float32) / np.float32(0.6)) for s in (0.02, 1.0, 3.0)]
            peaked = np.full(V, 1e-9, dtype=np.float32)
            peaked[:3] = [1.0, 0.5, 0.2]
            rows.append(peaked)
            worst = np.full(V, np.nextafter(cut_edge, np.float32(0)), dtype=np.float32)   # everything just below the cut
            worst[0] = 1.0
            rows.append(worst)
            for e in rows:
                e = (e / e.max()).astype(np.float32)         # the kernel's e = exp(x - max): the maximum is exactly 1
                m = fix(e)
                Z = int(m.sum())
                tau = int(np.floor(np.float64(np.float32(top_p)) * np.float64(Z)))
                cand = (m != 0) & (e.view(np.uint32) >= cutpat)
                assert int(m[cand].sum()) > tau, (V, top_p, int(m[cand].sum()), tau)
