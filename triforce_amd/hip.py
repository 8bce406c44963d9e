"""ctypes binding of libtriforce_hip.so (C ABI in include/triforce_hip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol is absent the
import of any op raises.  Build it with ``python -m triforce_amd.build`` (hipcc, gfx950).
"""
import ctypes
import os

from .build import LIB_PATH

_c = ctypes
_vp, _i32, _i64, _f32 = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

# name -> (restype, argtypes); mirrors include/triforce_hip.h one to one
SIGNATURES = {
    "tf_abi_version": (_i32, []),
    "tf_attn_decode_ws_floats": (_i64, [_i32, _i32, _i32, _i32]),
    "tf_attn_decode_pick_nsplit": (_i32, [_i32, _i32]),
    "tf_attn_decode": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _i32, _i32, _f32, _i32, _vp, _i64, _vp]),
    "tf_attn_decode_fused": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _i32, _i32, _f32, _i32, _vp, _i64, _vp,
                                    _vp]),
    "tf_attn_decode_act": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp, _i32, _i32, _f32, _i32, _vp,
                                  _i64, _vp, _vp]),
    "tf_attn_tune": (_i32, [_i32, _i32]),
    "tf_attn_block_ws_floats": (_i64, [_i32, _i32, _i32]),
    "tf_attn_block_pick_nsplit": (_i32, [_i32, _i32, _i32]),
    "tf_attn_prefill_pick_nsplit": (_i32, [_i32, _i32, _i32]),
    "tf_attn_prefill_ws_floats": (_i64, [_i32, _i32, _i32, _i32]),
    "tf_attn_prefill": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _i64, _vp]),
    "tf_attn_block": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _i64, _vp, _i32,
                             _i32, _i32, _vp]),
    "tf_attn_rope_on_read": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp]),
    "tf_retrieval_score": (_i32, [_vp, _i64, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tf_retrieval_topk": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "tf_retrieval_gather": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "tf_kv_copy_rows": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tf_kv_shift_rows": (_i32, [_vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tf_kv_copy_rows_pair": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32,
                                    _i32, _vp]),
    "tf_kv_shift_rows_pair": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tf_kv_gather_rows": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tf_tree_accept": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp]),
    "tf_sample_without_replacement": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "tf_rmsnorm": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "tf_rope_append": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tf_silu_mul": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "tf_embed_rows": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "tf_set_tokens": (_i32, [_vp, _i32, _vp, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _i32, _vp]),
    "tf_skinny_gemm": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "tf_skinny_gemm_ex": (_i32, [_vp, _vp, _i64, _vp, _f32, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "tf_skinny_gemm_swiglu_ex": (_i32, [_vp, _vp, _vp, _i64, _vp, _f32, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "tf_skinny_qkv_rope": (_i32, [_vp, _vp, _i64, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp,
                                  _i32, _i32, _i32, _i32, _i32, _vp]),
    "tf_skinny_gemm_swiglu": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "tf_skinny_gemm_act": (_i32, [_vp, _vp, _i64, _i64, _vp, _f32, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32,
                                  _i32, _i32, _vp]),
    "tf_skinny_gemm_swiglu_act": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _f32, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "tf_skinny_qkv_rope_act": (_i32, [_vp, _vp, _i64, _i64, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32,
                                      _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tf_skinny_gemm_swiglu_n8": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _f32, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "tf_skinny_qkv_rope_n8": (_i32, [_vp, _vp, _i64, _i64, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32,
                                     _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tf_sg_tune": (_i32, [_i32, _i32]),
    "tf_sg_workspace": (_i32, [_vp, _i64]),
    "tf_topp_probs": (_i32, [_vp, _vp, _i32, _i32, _f32, _f32, _vp]),
    "tf_sample_inverse_cdf": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "tf_accept_chain": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "tf_middle_accept": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "tf_mid_record_tokens": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "tf_sample_inverse_cdf_cur": (_i32, [_vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    "tf_middle_accept_cur": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "tf_accept_chain_step": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _vp, _i32,
                                    _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "tf_accept_chain_cur": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "tf_kv_h2d_async": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "tf_kv_d2h_async": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "tf_ar_flags_bytes": (_i32, []),
    "tf_ar_ipc_handle_bytes": (_i32, []),
    "tf_ar_alloc": (_i32, [_i64, _vp]),
    "tf_ar_free": (_i32, [_vp]),
    "tf_ar_get_ipc_handle": (_i32, [_vp, _vp]),
    "tf_ar_open_ipc_handle": (_i32, [_vp, _vp]),
    "tf_ar_close_ipc_handle": (_i32, [_vp]),
    "tf_allreduce_oneshot": (_i32, [_vp, _vp, _i32, _i32, _vp, _i64, _vp]),
    "tf_allreduce_oneshot_add": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp]),
    "tf_allreduce_oneshot_add_ss": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp]),
    "tf_allreduce_oneshot_alt": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp]),
    "tf_allreduce_oneshot_act": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp]),
    "tf_ar_litmus_stage": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "tf_ar_litmus_check": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "tf_xchg_ctl_bytes": (_i64, []),
    "tf_skinny_gemm_xchg": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i32, _i32, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp,
                                   _i32, _i32, _i32, _vp]),
    "tf_xchg_error": (_i32, [_vp]),
    "tf_xchg_set_error": (_i32, [_vp, _i32, _vp, _i32]),
    "tf_xchg_tune": (_i32, [_i32, _i32]),
    "tf_xchg_reset": (_i32, [_vp]),
    "tf_ar_error": (_i32, [_vp]),
    "tf_ar_epoch": (_i64, [_vp]),
    "tf_ar_set_error_mirror": (_i32, [_vp, _vp]),
    "tf_ar_inject_error": (_i32, [_vp, _i32]),
}

TF_DRAFT_MAX_LAYERS = 8


class TfDraftModel(_c.Structure):
    """include/triforce_hip.h: TfDraftModel (device pointers of the 68M draft model's weights)."""
    _fields_ = ([("embed", _vp)] + [(n, _vp * TF_DRAFT_MAX_LAYERS) for n in
                                    ("ln1", "wqkv", "wo", "ln2", "wgate", "wup", "wdown")]
                + [("norm", _vp), ("lm_head", _vp), ("cos", _vp), ("sin", _vp)]
                + [(n, _i32) for n in ("layers", "hidden", "heads", "head_dim", "inter", "vocab")]
                + [("eps", _f32), ("scale", _f32)])


class TfDraftCache(_c.Structure):
    _fields_ = [("k", _vp * TF_DRAFT_MAX_LAYERS), ("v", _vp * TF_DRAFT_MAX_LAYERS), ("stride_t", _i64), ("stride_h", _i64)]


SIGNATURES["tf_draft_forward_ws_bytes"] = (_i64, [_c.POINTER(TfDraftModel), _i32])
SIGNATURES["tf_draft_forward_68m"] = (_i32, [_c.POINTER(TfDraftModel), _c.POINTER(TfDraftCache), _vp, _i32, _i32, _i32, _vp,
                                             _vp, _f32, _f32, _vp, _i64, _vp])
SIGNATURES["tf_topp_multi_ctl_bytes"] = (_i64, [])
SIGNATURES["tf_topp_multi_ws_bytes"] = (_i64, [_i32, _i32])
SIGNATURES["tf_topp_probs_multi"] = (_i32, [_vp, _vp, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _i64, _vp])
SIGNATURES["tf_topp_multi_tune"] = (_i32, [_i32, _i32])
SIGNATURES["tf_topp_multi_error"] = (_i32, [_vp])
SIGNATURES["tf_topp_multi_reset"] = (_i32, [_vp])
SIGNATURES["tf_draft_persist_ctl_bytes"] = (_i64, [])
SIGNATURES["tf_draft_persist_ws_bytes"] = (_i64, [_c.POINTER(TfDraftModel)])
SIGNATURES["tf_draft_persist_supported"] = (_i32, [_c.POINTER(TfDraftModel), _i32, _i32])
SIGNATURES["tf_draft_forward_68m_persist"] = (_i32, [_c.POINTER(TfDraftModel), _c.POINTER(TfDraftCache), _vp, _i32, _i32, _i32,
                                                     _vp, _vp, _f32, _f32, _vp, _i64, _vp, _vp])
SIGNATURES["tf_draft_persist_tune"] = (_i32, [_i32, _i32])
SIGNATURES["tf_draft_persist_stamps"] = (_i32, [_vp])
SIGNATURES["tf_draft_persist_error"] = (_i32, [_vp])
SIGNATURES["tf_draft_persist_reset"] = (_i32, [_vp, _vp, _i32])

ABI_VERSION = 1
_lib = None


class TriforceHipError(RuntimeError):
    pass


def _preload_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so.7.  Our library must bind to THAT runtime instance (the one
    that owns torch's device context and streams): load it first so the dynamic linker reuses it for our
    NEEDED libamdhip64.so.7 instead of opening the system copy as a second, uninitialised runtime
    (symptom: every launch fails with hipErrorNoDevice)."""
    import torch  # noqa: F401  (imports libtorch_hip -> torch/lib/libamdhip64.so)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def lib():
    """The loaded library; raises (never falls back) when it cannot be loaded."""
    global _lib
    if _lib is None:
        _preload_torch_hip_runtime()
        path = os.environ.get("TRIFORCE_HIP_LIB", LIB_PATH)        # tuning builds (triforce_amd.build.build_variant)
        if not os.path.exists(path):
            raise TriforceHipError(
                f"{path} not found: the HIP extension is required (python -m triforce_amd.build); "
                "there is no CPU fallback")
        L = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise TriforceHipError(f"{path} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        if L.tf_abi_version() != ABI_VERSION:
            raise TriforceHipError(f"ABI mismatch: library {L.tf_abi_version()} != binding {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        kind = {-22: "TF_EINVAL", -28: "TF_ENOSPC", -34: "TF_ERANGE"}.get(rc, f"hipError {rc}" if rc > 0 else str(rc))
        raise TriforceHipError(f"{what} failed: {kind}")
