"""The C-ABI library loads, exports every symbol include/triforce_hip.h declares, and rejects bad
arguments before launching anything (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "triforce_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from triforce_amd import hip
    lib = hip.lib()
    names = _header_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"libtriforce_hip.so does not export {n}"
    assert set(names) == set(hip.SIGNATURES), "ctypes binding and header disagree"
    assert lib.tf_abi_version() == hip.ABI_VERSION


def test_host_side_helpers():
    from triforce_amd import hip
    lib = hip.lib()
    # one workgroup per CU (nsplit * H ~ 256) while every workgroup keeps >= 8 key tiles (profiles/r02_nsplit_sweep.json)
    assert lib.tf_attn_decode_pick_nsplit(32, 124928) == 8
    assert lib.tf_attn_decode_pick_nsplit(32, 4103) == 8
    assert lib.tf_attn_decode_pick_nsplit(16, 130066) == 16
    assert lib.tf_attn_decode_pick_nsplit(4, 4103) == 32           # 257 tiles / 8
    assert lib.tf_attn_decode_pick_nsplit(4, 130066) == 64
    assert lib.tf_attn_decode_pick_nsplit(1, 130066) == 128         # capped by the merge kernel's split limit
    assert lib.tf_attn_decode_pick_nsplit(2, 16) == 1
    assert lib.tf_attn_decode_ws_floats(32, 8, 128, 32) == 32 * 32 * 16 * 130
    assert lib.tf_attn_decode_ws_floats(16, 18, 128, 4) == 16 * 4 * 32 * 130


def test_prefill_split_rule_properties():
    """tf_attn_prefill_pick_nsplit for every head count a TP shard can have: (split, head) pairs divide over the 8 XCDs,
    the count stays within the merge kernel's limit, short caches are not split, and the workspace query matches the
    kernel's per-row-block layout."""
    from triforce_amd import hip
    lib = hip.lib()
    for H in list(range(1, 41)) + [64]:
        for sq in (129, 1024, 2048, 4096):
            for sk in (sq, 8192, 130048):
                n = lib.tf_attn_prefill_pick_nsplit(H, sq, sk)
                assert 1 <= n <= 128 and (H * n) % 8 == 0, (H, sq, sk, n)
                nrb = (sq + 127) // 128
                assert lib.tf_attn_prefill_ws_floats(H, sq, 128, n) == nrb * H * n * 128 * 130
    assert lib.tf_attn_prefill_pick_nsplit(32, 1024, 1024) == 1
    assert lib.tf_attn_prefill_pick_nsplit(8, 1024, 130048) >= lib.tf_attn_prefill_pick_nsplit(32, 1024, 130048)


def test_bad_arguments_are_rejected_without_launching():
    from triforce_amd import hip
    lib = hip.lib()
    null = ctypes.c_void_p(0)
    assert lib.tf_attn_decode(null, null, null, null, 128, 128, 1, 1, null, 1, 128, 1.0, 1, null, 0, null) == -22
    m = hip.TfDraftModel()
    m.layers, m.hidden, m.heads, m.head_dim, m.inter, m.vocab = 2, 768, 12, 64, 3072, 32000
    assert lib.tf_draft_forward_ws_bytes(ctypes.byref(m), 7) == 3 * 10752 + 256 + 43008 + 6144     # 256-byte aligned pieces
    assert lib.tf_draft_forward_ws_bytes(None, 7) == 0
    assert lib.tf_draft_forward_68m(ctypes.byref(m), None, null, 7, 0, 7, null, null, 0.6, 0.9, null, 0, null) == -22
    # the one-launch form: control block / workspace sizes, shape gate (the device query is not reached for a refused shape)
    assert lib.tf_draft_persist_ctl_bytes() == 64 + 12 * 9 * 64 + 12 * 8 * 64      # head, counters, READY flags
    assert lib.tf_draft_persist_ws_bytes(ctypes.byref(m)) == 3 * 24576 + 98304 + 6144 + 1024 + 256 * 128 * 8 + 2048 + 1024    # ..., candidates, sums, counts
    assert lib.tf_draft_persist_ws_bytes(None) == 0
    assert lib.tf_draft_persist_supported(ctypes.byref(m), 17, 100) == -22 and lib.tf_draft_persist_supported(None, 1, 1) == -22
    m.inter = 2048
    assert lib.tf_draft_persist_supported(ctypes.byref(m), 4, 100) == -22
    m.inter = 3072
    assert lib.tf_draft_forward_68m_persist(ctypes.byref(m), None, null, 7, 0, 7, null, null, 0.6, 0.9, null, 0, null, null) == -22
    assert lib.tf_draft_persist_error(null) == -22 and lib.tf_draft_persist_reset(null, null, 0) == -22
    assert lib.tf_draft_persist_tune(9, 1) == -1
    # the multi-workgroup top-p: sizes, shape gate
    assert lib.tf_topp_multi_ctl_bytes() == 64 + 2 * 2 * 32 * 64
    assert lib.tf_topp_multi_ws_bytes(7, 32000) == 512 + 1024 + 512 + 7 * 16 * 500 * 4 * 8
    assert lib.tf_topp_multi_ws_bytes(33, 32000) == 0 and lib.tf_topp_multi_ws_bytes(4, 32002) == 0
    assert lib.tf_topp_probs_multi(null, null, null, 7, 32000, 0.6, 0.9, null, null, 0, null) == -22
    assert lib.tf_topp_multi_error(null) == -22 and lib.tf_topp_multi_reset(null) == -22 and lib.tf_topp_multi_tune(5, 1) == -1
    assert lib.tf_attn_prefill_pick_nsplit(32, 1024, 124928) == 4          # 1024 / (32 heads x 8 row blocks)
    assert lib.tf_attn_prefill_pick_nsplit(32, 1024, 1024) == 1            # 16 slabs: no split
    assert (12 * lib.tf_attn_prefill_pick_nsplit(12, 256, 4096)) % 8 == 0  # pairs are dealt to the 8 XCDs
    assert lib.tf_attn_prefill_ws_floats(32, 1000, 128, 4) == 8 * 32 * 4 * 128 * 130
    assert lib.tf_attn_decode_fused(null, null, null, null, 128, 128, 1, 1, null, 1, 128, 1.0, 1, null, 0, null, null) == -22
    assert lib.tf_rmsnorm(null, null, null, null, null, 1, 8, 1e-6, null) == -22
    assert lib.tf_retrieval_topk(null, null, 10, 2, 1, null) == -22
    assert lib.tf_kv_copy_rows(null, 0, 0, 0, null, 0, 0, 0, 0, 0, 0, 1, 1, 8, null) == 0      # n == 0 is a no-op
    with pytest.raises(hip.TriforceHipError):
        hip.check(-22, "x")


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product ops raise on non-device tensors instead of computing."""
    import torch
    from triforce_amd import hip, ops
    x = torch.zeros(2, 8, dtype=torch.float16)
    with pytest.raises(hip.TriforceHipError):
        ops.rmsnorm(x, torch.ones(8, dtype=torch.float16), 1e-6)
    with pytest.raises(hip.TriforceHipError):
        ops.silu_mul(torch.zeros(2, 16, dtype=torch.float16))


_QUERIES = {"tf_abi_version", "tf_attn_block_pick_nsplit", "tf_attn_block_ws_floats", "tf_attn_decode_pick_nsplit",
            "tf_attn_prefill_pick_nsplit", "tf_attn_prefill_ws_floats", "tf_draft_forward_ws_bytes",
            "tf_attn_decode_ws_floats", "tf_ar_flags_bytes", "tf_ar_ipc_handle_bytes",
            "tf_sg_tune", "tf_sg_workspace", "tf_xchg_ctl_bytes", "tf_attn_tune", "tf_xchg_tune",
            "tf_draft_persist_ctl_bytes", "tf_draft_persist_ws_bytes", "tf_draft_persist_tune", "tf_draft_persist_stamps",
            "tf_topp_multi_ctl_bytes", "tf_topp_multi_ws_bytes", "tf_topp_multi_tune"}   # launch-rule knob / workspace registration: nothing launched (NULL = remove)


@pytest.mark.parametrize("fill", [1, 8, -1])
def test_every_entry_point_rejects_null_buffers(fill):
    """Boundary contract (include/triforce_hip.h): every launching entry point validates its arguments BEFORE touching
    the device and returns TF_EINVAL — with all pointers NULL and all sizes = `fill` none may launch, crash or report
    success (tf_kv_shift_rows with src == dst is the one documented no-op)."""
    from triforce_amd import hip
    lib = hip.lib()
    for name, (_, argtypes) in sorted(hip.SIGNATURES.items()):
        if name in _QUERIES:
            continue
        args = []
        for t in argtypes:
            if t is ctypes.c_void_p:
                args.append(ctypes.c_void_p(0))
            elif isinstance(t, type) and issubclass(t, ctypes._Pointer):
                args.append(None)                                       # struct pointers (TfDraftModel*, ...): NULL
            elif t in (ctypes.c_float, ctypes.c_double):
                args.append(1.0)
            else:
                args.append(t(fill))
        rc = getattr(lib, name)(*args)
        if name in ("tf_kv_shift_rows", "tf_kv_shift_rows_pair"):
            assert rc == 0
        else:
            assert rc == -22, f"{name}(NULL..., sizes={fill}) returned {rc}"
