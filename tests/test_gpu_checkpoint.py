"""The checkpoint path on the device (reference test/on_chip.py:48-56): a seeded tiny model written as `config.json` (model-card
rope_scaling dict) + `*.safetensors` and loaded through `LlamaForCausalLM.from_pretrained` must give logits BIT-IDENTICAL to
the `from_state_dict` construction every other parity test uses — target class and 68M class — and the entry script itself
must run end to end from such directories (`test/on_chip.py --weights <dir> --draft-weights <dir>`), emitting through the
HIP library.  (CPU half: tests/test_checkpoint_cpu.py.)"""
import os
import re
import subprocess
import sys

import pytest
import torch

from tests.test_checkpoint_cpu import tiny_cfgs, write_checkpoint

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_from_pretrained_logits_equal_from_state_dict_on_device(tmp_path):
    from triforce_amd.models.cache import FlashSimpleCache, StreamingLLMEvictionCache
    from triforce_amd.models.config_yarn import LlamaConfig
    from triforce_amd.models.modeling_llama import LlamaForCausalLM
    from triforce_amd.models.modeling_llama_68m import LlamaForCausalLM as Draft
    tcfg, dcfg = tiny_cfgs()
    ids = torch.randint(3, tcfg["vocab_size"], (1, 24), generator=torch.Generator().manual_seed(3)).to(DEV)
    tpath, dpath = str(tmp_path / "target"), str(tmp_path / "draft")
    tsd, dsd = write_checkpoint(tpath, tcfg, 31, shards=2), write_checkpoint(dpath, dcfg, 32)
    outs = []
    for model in (LlamaForCausalLM.from_pretrained(tpath, torch_dtype=torch.float16, device_map=DEV),
                  LlamaForCausalLM.from_state_dict(LlamaConfig.from_dict(tcfg), tsd, DEV)):
        cache = FlashSimpleCache(model, 64)
        a = model(input_ids=ids[:, :17], kv_cache=cache, graph_cache=None).logits       # a prefill block ...
        b = model(input_ids=ids[:, 17:], kv_cache=cache, graph_cache=None).logits       # ... and a 7-row decode block
        outs.append((a, b))
    for x, y in zip(*outs):
        assert torch.isfinite(x).all() and torch.equal(x, y)
    outs = []
    for model in (Draft.from_pretrained(dpath, torch_dtype=torch.float16, device_map=DEV),
                  Draft.from_state_dict(LlamaConfig.from_dict(dcfg), dsd, DEV)):
        cache = StreamingLLMEvictionCache(model, start_size=16, recent_size=256 - 16 - 6, gamma=6)
        outs.append(model(input_ids=ids[:, :20], kv_cache=cache, graph_cache=None).logits)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_on_chip_script_runs_from_checkpoint_directories(tmp_path):
    tcfg, dcfg = tiny_cfgs()
    tpath, dpath = str(tmp_path / "target"), str(tmp_path / "draft")
    write_checkpoint(tpath, tcfg, 41)
    write_checkpoint(dpath, dcfg, 42)
    cmd = [sys.executable, os.path.join(ROOT, "test", "on_chip.py"), "--target", "tiny", "--weights", tpath, "--draft-weights", dpath,
           "--prefill", "1024", "--budget", "256", "--chunk_size", "8", "--gamma", "4", "--gen_len", "24", "--greedy"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = r.stdout
    assert tpath in out, "the printed configuration does not name the checkpoint directory"
    m = re.search(r"\[E2E Speedup\]: ([0-9.]+)", out)
    assert m and float(m.group(1)) > 0, out[-1500:]
    acc = re.search(r"average acceptance rate \(NOT per token\): ([0-9.]+)", out)
    assert acc and 0.0 <= float(acc.group(1)) <= 1.0
