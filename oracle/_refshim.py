"""Import the *unmodified* reference tree (/root/reference) on a CPU-only box.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/gen_golden.py`` (and nothing else) to run the
reference's own Python on seeded tiny models so that (a) the CPU restatement in
``oracle/ref_ops.py`` / ``oracle/ref_model.py`` can be pinned against it and (b) golden
vectors can be written to ``tests/golden/``.  ``/root/reference`` does not exist on the
GPU box, so nothing under ``tests/``, ``bench.py`` or ``__graft_entry__`` imports this file.

Four runtime shims are needed; none of them edits reference source:
  1. ``flash_attn.flash_attn_with_kvcache`` -> eager fp32 attention, bottom-right causal
     (flash-attn >= 2.1 semantics; call sites modeling_llama.py:240,
     modeling_llama_68m.py:186, tensor_op.py:166,168,316).  Installed AFTER transformers is
     imported because transformers 5.x probes ``find_spec('flash_attn')``.
  2. ``termcolor.colored`` -> identity (utils/misc.py:2, test/on_chip.py:10).
  3. ``torch.Tensor.cuda`` -> identity (models/cache.py:154,166,172 call ``.cuda()``).
  4. ``models.modeling_llama.apply_rotary_pos_emb`` -> the 4.37-style copy the reference
     itself carries in models/tensor_op.py:25-50 (transformers >= 4.38 dropped ``position_ids``).
"""
import importlib
import math
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def eager_flash_attn_with_kvcache(q, k_cache, v_cache, softmax_scale=None, causal=False, cache_seqlens=None):
    """fp32 restatement of flash_attn_with_kvcache for bsz==1 MHA: q (b,sq,h,d), caches (b,sk,h,d).

    Bottom-right aligned causal mask: query i attends keys [0, sk - sq + i].
    """
    b, sq, h, d = q.shape
    sk = k_cache.shape[1]
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(d)
    scale = float(softmax_scale)
    qf = q.float().permute(0, 2, 1, 3)            # b h sq d
    kf = k_cache.float().permute(0, 2, 3, 1)      # b h d sk
    vf = v_cache.float().permute(0, 2, 1, 3)      # b h sk d
    s = torch.matmul(qf, kf) * scale
    if causal:
        qi = torch.arange(sq).view(sq, 1)
        kj = torch.arange(sk).view(1, sk)
        s = s.masked_fill(kj > (sk - sq + qi), float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vf)                        # b h sq d
    return o.permute(0, 2, 1, 3).to(q.dtype).contiguous()


_loaded = {}


def load_reference():
    """Returns a namespace of reference modules, importing them once with the shims installed."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)

    import transformers  # noqa: F401  (must precede the flash_attn stub)
    import transformers.models.llama.modeling_llama  # noqa: F401

    if "termcolor" not in sys.modules:
        tc = types.ModuleType("termcolor")
        tc.colored = lambda s, *a, **k: s
        sys.modules["termcolor"] = tc

    fa = types.ModuleType("flash_attn")
    fa.flash_attn_with_kvcache = eager_flash_attn_with_kvcache
    fa.__spec__ = importlib.machinery.ModuleSpec("flash_attn", None)
    sys.modules["flash_attn"] = fa

    torch.Tensor.cuda = lambda self, *a, **k: self

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    # transformers>=4.38 has a 4-arg apply_rotary_pos_emb; the reference calls the 4.37 5-arg form.
    tensor_op = importlib.import_module("models.tensor_op")
    import transformers.models.llama.modeling_llama as hf_llama
    hf_llama.apply_rotary_pos_emb = tensor_op.apply_rotary_pos_emb
    if not hasattr(hf_llama, "repeat_kv"):
        hf_llama.repeat_kv = tensor_op.repeat_kv

    mods = dict(
        tensor_op=tensor_op,
        cache=importlib.import_module("models.cache"),
        config_yarn=importlib.import_module("models.config_yarn"),
        modeling_llama=importlib.import_module("models.modeling_llama"),
        modeling_llama_68m=importlib.import_module("models.modeling_llama_68m"),
        sampling=importlib.import_module("utils.sampling"),
        graph_infer=importlib.import_module("utils.graph_infer"),
        decoding=importlib.import_module("utils.decoding"),
    )
    mods["modeling_llama"].apply_rotary_pos_emb = tensor_op.apply_rotary_pos_emb
    _loaded.update(mods)
    return types.SimpleNamespace(**_loaded)


class EagerEngine:
    """CPU stand-in with the GraphInferenceEngine surface (utils/graph_infer.py:129-194):
    the captured-graph entry points are routed to the eager InferenceEngine methods they capture."""

    def __init__(self, ref, model, cache, graph_cache, draft, draft_cache, temperature, top_p):
        self.engine = ref.graph_infer.InferenceEngine(model, cache, graph_cache, draft, draft_cache)
        self.temperature = temperature
        self.top_p = top_p

    def graph_draft_inference(self, input_ids, gamma_offset=0):
        return self.engine.draft_run(input_ids=input_ids, gamma_offset=gamma_offset, probs=True,
                                     temperature=self.temperature, top_p=self.top_p).clone()

    def graph_draft_prefill(self, input_ids):
        return self.engine.draft_run(input_ids=input_ids)

    def inference(self, input_ids):
        return self.engine.model_run(input_ids=input_ids)

    def graph_verify(self, input_ids, position_ids):
        return self.engine.model_verify(input_ids=input_ids, position_ids=position_ids, probs=True,
                                        temperature=self.temperature, top_p=self.top_p).clone()

    def update_graph_cache(self):
        self.engine.graph_cache.update_graph_cache(kv_cache=self.engine.kv_cache)


class FakeTokenizer:
    eos_token_id = 2

    def decode(self, ids, **kw):
        return ""


# ----------------------------------------------------------------------------------------------
# Sequoia tree path: models/TP_llama_tree.py + utils/SpecTree_TP.py on a CPU-only box
# ----------------------------------------------------------------------------------------------
class _CudaProxy:
    """Stands in for ``torch.cuda`` inside the reference's TP modules: streams / synchronize become no-ops."""

    class Stream:
        def __init__(self, *a, **k):
            pass

    @staticmethod
    def stream(_s):
        import contextlib
        return contextlib.nullcontext()

    @staticmethod
    def synchronize(*a, **k):
        return None

    @staticmethod
    def set_device(*a, **k):
        return None

    def __getattr__(self, name):
        return getattr(torch.cuda, name)


class _TorchProxy:
    """Module-level ``torch`` replacement for ONE reference module: ``torch.device("cuda", r)`` -> cpu,
    ``torch.cuda`` -> _CudaProxy, ``pin_memory=True`` dropped from factory calls; everything else forwards."""

    def __init__(self):
        self.cuda = _CudaProxy()

    def device(self, *a, **k):
        return torch.device("cpu")

    def zeros(self, *a, **k):
        k.pop("pin_memory", None)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        return torch.zeros(*a, **k)

    def __getattr__(self, name):
        return getattr(torch, name)


def load_reference_tree():
    """Reference modules of the Sequoia path, importable on CPU: a 1-rank gloo group (the reference calls
    dist.all_reduce / broadcast / barrier unconditionally), Tensor.pin_memory -> identity, and a torch proxy in
    the globals of models.cache / models.TP_layers / models.TP_llama_tree.  No reference source is edited."""
    ref = load_reference()
    if "tree" in _loaded:
        return types.SimpleNamespace(**_loaded)
    import torch.distributed as dist
    if not dist.is_initialized():
        import os
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        dist.init_process_group("gloo", rank=0, world_size=1)
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    proxy = _TorchProxy()
    ref.cache.torch = proxy
    tp_layers = importlib.import_module("models.TP_layers")
    tp_layers.torch = proxy
    tree = importlib.import_module("models.TP_llama_tree")
    tree.torch = proxy
    spectree = importlib.import_module("utils.SpecTree_TP")
    _loaded.update(tree=tree, spectree=spectree, tp_layers=tp_layers)
    return types.SimpleNamespace(**_loaded)


def load_reference_tp():
    """Reference modules of the tensor-parallel chain path (models/TP_llama.py + utils/decoding.py *_Dist) importable
    on CPU, with the same runtime shims as load_reference_tree: 1-rank gloo group, pin_memory -> identity, a torch
    proxy in the globals of models.cache / models.TP_layers / models.TP_llama.  No reference source is edited."""
    ref = load_reference_tree()                 # group + proxies for cache / TP_layers
    if "tp" in _loaded:
        return types.SimpleNamespace(**_loaded)
    tp = importlib.import_module("models.TP_llama")
    tp.torch = _TorchProxy()
    _loaded.update(tp=tp)
    return types.SimpleNamespace(**_loaded)
