"""Tensor-level wrappers over the C ABI (include/triforce_hip.h).

PyTorch here is plumbing only: it owns device memory and the stream; every function below
validates its tensors, takes raw device pointers and enqueues hand-written gfx950 kernels on
torch's current HIP stream (so they are captured by ``torch.cuda.graph`` like any torch op).
No CPU path exists: tensors that are not on a HIP device raise.

KV "layer views" are 3-D tensors (H, T, D) with unit stride on D and arbitrary H/T strides, so
the physical layout (head-major [L][H][T][D] in triforce_amd, token-major in the reference)
is the caller's choice.  Other modules must call these as ``ops.<name>(...)``.
"""
import ctypes
import functools

import torch
import torch.nn.functional as F

from . import hip

_HALF = torch.float16


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise hip.TriforceHipError("triforce_amd ops need HIP device tensors (no CPU fallback)")


def _dev_or_pinned(t):
    """Small result records may live in pinned host memory (device-writable under unified addressing): the host then
    polls the record instead of copying it back."""
    if not (t.is_cuda or t.is_pinned()):
        raise hip.TriforceHipError("result record must be a HIP device tensor or pinned host memory")


def _kv(t):
    assert t.dim() == 3 and t.stride(2) == 1 and t.dtype == _HALF, "KV layer view must be (H,T,D) fp16, D contiguous"
    return t.stride(1), t.stride(0)          # stride_t, stride_h


# ------------------------------------------------------------------------------------------
SKINNY_MAX_ROWS = 32
# Decode-layer fusion level (A/B switch, env TRIFORCE_FUSE): "all" = norm prologues fed by the GEMM-to-GEMM
# sum-of-squares hand-off + residual / RoPE epilogues (6 launches per layer); "all2" = same but every norm prologue
# re-reads x; "rope" = only the RoPE + KV-append epilogue; "none" = one launch per op (9 per layer).
import os as _os
FUSE_MODE = _os.environ.get("TRIFORCE_FUSE", "all")


def pack_weight(w):
    """[N, K] fp16 -> MFMA-operand order [N/16][K/32][4 (g)][16 (i)][8]: one 16x32 tile = one contiguous KiB."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0 and w.dtype == _HALF
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def rope_row_order(H, D, device=None):
    """Row permutation of a fused [q | k | v] weight for the RoPE epilogue of tf_skinny_qkv_rope: every 16-row panel
    of the q and k sections holds rows d0..d0+7 and their rotary partners d0+D/2..d0+D/2+7 of one head; v rows keep
    their order."""
    assert D % 32 == 0
    d = torch.arange(D // 2, device=device).view(D // 16, 8)
    per_head = torch.cat([d, d + D // 2], dim=1).reshape(-1)                     # (D,)
    qk = (torch.arange(2 * H, device=device).view(-1, 1) * D + per_head.view(1, -1)).reshape(-1)
    return torch.cat([qk, torch.arange(2 * H * D, 3 * H * D, device=device)])


# ---- narrow panels (round 5; csrc/gemv.hip skinny_gemm_n8_kernel) ----------------------------------------------------
# The q|k|v / gate|up shards of a tensor-parallel rank have 86-120 16-row panels: one workgroup per panel leaves 2/3 of the
# 256 CUs idle and each workgroup is bound by what one CU can pull.  Weights with at most N8_MAX_PANELS 16-row panels per
# weight stream get a SECOND packed copy in 8-row panels (twice the workgroups, half the bytes each, no hand-off) that the
# two norm GEMMs use for blocks of <= 24 rows.  TRIFORCE_GEMM_N8 = 0 disables, TRIFORCE_GEMM_N8_MAX_PANELS sets the limit.
N8_ENABLED = _os.environ.get("TRIFORCE_GEMM_N8", "1") != "0"
N8_MAX_PANELS = int(_os.environ.get("TRIFORCE_GEMM_N8_MAX_PANELS", "128"))
N8_MAX_ROWS = 24
N8_MIN_K = 1024


def pack_weight_n8(w):
    """[N, K] fp16 -> narrow-panel operand order [N/8][K/64][4 (g)][2 (chunk parity)][8 (row)][8]: per 8-row panel and
    64-wide super-chunk one contiguous KiB whose 64 16-byte pieces are the A operand of ONE 16x16x32 MFMA — rows 0-7 of
    the operand = the panel's rows over the even 32-wide chunk, rows 8-15 = the same rows over the odd chunk."""
    N, K = w.shape
    assert N % 8 == 0 and K % 64 == 0 and w.dtype == _HALF
    return w.view(N // 8, 8, K // 64, 2, 4, 8).permute(0, 2, 4, 3, 1, 5).contiguous()


def rope_row_order_n8(H, D, device=None):
    """rope_row_order for 8-row panels: every panel of the q and k sections holds rows d0..d0+3 and their rotary partners
    d0+D/2..d0+D/2+3 of one head; v rows keep their order."""
    assert D % 32 == 0
    d = torch.arange(D // 2, device=device).view(D // 8, 4)
    per_head = torch.cat([d, d + D // 2], dim=1).reshape(-1)                     # (D,)
    qk = (torch.arange(2 * H, device=device).view(-1, 1) * D + per_head.view(1, -1)).reshape(-1)
    return torch.cat([qk, torch.arange(2 * H * D, 3 * H * D, device=device)])


def n8_applies(N_stream, K):
    """Does a weight stream of N_stream rows x K get (and use) the narrow-panel copy?"""
    return (N8_ENABLED and N_stream % 16 == 0 and N_stream // 16 <= N8_MAX_PANELS and K % 64 == 0 and K >= N8_MIN_K)


# Split-K workspace of the skinny GEMM (csrc/gemv.hip SgKsplit, tf_sg_workspace): one zero-filled 8 MiB block per device,
# registered when the first weight is packed there (i.e. before any hipGraph capture) and never freed.  Few-panel GEMMs
# (the q|k|v / gate|up shards of a tensor-parallel rank) can then split K across up to 4 workgroups per panel —
# TRIFORCE_GEMM_KSPLIT=1; off by default: measured without gain in situ (profiles/r04_tp_shard_structural_ab.jsonl).
_SG_WS = {}
_SG_WS_BYTES = 8 << 20


def _ensure_sg_workspace(device):
    device = torch.device(device)
    if device.type != "cuda" or device in _SG_WS:
        return
    with torch.cuda.device(device):
        ws = torch.zeros(_SG_WS_BYTES, dtype=torch.uint8, device=device)
        torch.cuda.synchronize(device)
        hip.check(hip.lib().tf_sg_workspace(_ptr(ws), ws.numel()), "tf_sg_workspace")
        if _os.environ.get("TRIFORCE_GEMM_KSPLIT", "0") == "1":          # measured: no gain (csrc/gemv.hip) — opt-in
            hip.lib().tf_sg_tune(3, 200)
        if "TRIFORCE_GEMM_DEEP_PANELS" in _os.environ:                   # A/B: largest panel count that keeps 2x the weights in flight
            hip.lib().tf_sg_tune(6, int(_os.environ["TRIFORCE_GEMM_DEEP_PANELS"]))
        if "TRIFORCE_GEMM_P2_GROUPS" in _os.environ:                     # A/B: two panels per wave while panels / 2 >= this
            hip.lib().tf_sg_tune(2, int(_os.environ["TRIFORCE_GEMM_P2_GROUPS"]))
        if "TRIFORCE_GEMM_P2_WAVES" in _os.environ:                      # A/B: waves per workgroup of that form (4 | 8)
            hip.lib().tf_sg_tune(1, int(_os.environ["TRIFORCE_GEMM_P2_WAVES"]))
        if "TRIFORCE_GEMM_N8_U" in _os.environ:                          # A/B: super-chunks per batch of the narrow-panel form (5 | 8)
            hip.lib().tf_sg_tune(7, int(_os.environ["TRIFORCE_GEMM_N8_U"]))
        if "TRIFORCE_GEMM_FEW_PANELS" in _os.environ:                    # A/B: largest panel count that runs 16 waves per panel
            hip.lib().tf_sg_tune(5, int(_os.environ["TRIFORCE_GEMM_FEW_PANELS"]))
    _SG_WS[device] = ws


class PackedLinear:
    """A weight matrix with (on a HIP device) its pre-packed copy for the skinny decode GEMM.  ``w`` stays
    available for the >32-row prefill GEMMs (hipBLASLt).  ``split`` = number of equal row blocks that were
    fused (2 for gate|up) — each block is packed on its own so the SwiGLU kernel can pair them.  ``rope=(H, D)``
    marks a fused q|k|v weight: a second packed copy in rotary-pair row order feeds tf_skinny_qkv_rope."""

    def __init__(self, w, split=1, pack=None, rope=None):
        self.w = w
        self.N, self.K = w.shape
        self.split = split
        pack = w.is_cuda if pack is None else pack
        ok = pack and (self.N // split) % 16 == 0 and self.K % 32 == 0
        if ok and w.is_cuda:
            _ensure_sg_workspace(w.device)
        self.parts = [pack_weight(b) for b in w.chunk(split, dim=0)] if ok else None
        self.wp = self.parts[0] if (ok and split == 1) else None
        self.rope = rope
        self.wp_rope = None
        if ok and rope is not None and rope[1] % 32 == 0 and self.N == 3 * rope[0] * rope[1]:
            self.wp_rope = pack_weight(w[rope_row_order(rope[0], rope[1], w.device)])
        # narrow-panel copies (few-panel shards only): gate|up streams, and the q|k|v weight in its 8-row rotary order
        self.parts_n8 = self.wp_rope_n8 = None
        if ok and w.is_cuda and split == 2 and n8_applies(self.N // split, self.K):
            self.parts_n8 = [pack_weight_n8(b) for b in w.chunk(split, dim=0)]
        if self.wp_rope is not None and w.is_cuda and n8_applies(self.N, self.K):
            self.wp_rope_n8 = pack_weight_n8(w[rope_row_order_n8(rope[0], rope[1], w.device)])

    def refresh_(self):
        """Re-pack IN PLACE after ``self.w`` was modified in place (captured hipGraphs keep their pointers)."""
        if self.parts is not None:
            for dst, blk in zip(self.parts, self.w.chunk(self.split, dim=0)):
                dst.copy_(pack_weight(blk))
        if self.wp_rope is not None:
            self.wp_rope.copy_(pack_weight(self.w[rope_row_order(self.rope[0], self.rope[1], self.w.device)]))
        if self.parts_n8 is not None:
            for dst, blk in zip(self.parts_n8, self.w.chunk(self.split, dim=0)):
                dst.copy_(pack_weight_n8(blk))
        if self.wp_rope_n8 is not None:
            self.wp_rope_n8.copy_(pack_weight_n8(self.w[rope_row_order_n8(self.rope[0], self.rope[1], self.w.device)]))
        return self


def _w(w):
    return w.w if isinstance(w, PackedLinear) else w


# ---- activation layouts of the decode path (include/triforce_hip.h, tf_skinny_gemm_act) -------------------------------
# Element (m, k) of an activation block lives at base[m * sm + (k // 8) * sk + k % 8].  The reference's tensors are
# row-major (sm = row stride, sk = 8).  The decode layer of this package keeps its residual stream, attention output and
# SwiGLU output K-OCTET-MAJOR (sm = 8, sk = 8 * rows): the B operand of a 16-row MFMA tile is then 4 runs of 256 bytes
# instead of 16 row fragments of 64 bytes — the 17..32-row GEMMs of the gamma = 16 verifies were bound by exactly those
# fragments (profiles/r03_gemm_rows_ab.jsonl: 13B q|k|v 29.7 us at 8 rows, 44.0 at 17; k-octet-major: 27.6 at 17,
# profiles/r04_gemm_layout_ab.jsonl).  TRIFORCE_ACT_LAYOUT = auto (default: see act_packed) | packed | rows.
ACT_LAYOUT = _os.environ.get("TRIFORCE_ACT_LAYOUT", "auto")
# auto: from 17 rows (two MFMA row tiles) — measured in situ (profiles/r04_act_layout_in_situ.jsonl): 13B gamma = 16
# retrieval verify 9 198 -> 8 319 us, target verify 25 640 -> 24 578; 7B gamma = 16 retrieval verify 5 287 -> 5 024; at
# 7 rows 3 345 -> 3 303 (1 %) and the tensor-parallel shard's eager 1-row step LOSES (host cost of the Act wrappers)
ACT_PACKED_MIN_ROWS = int(_os.environ.get("TRIFORCE_ACT_PACKED_MIN_ROWS", "17"))


def act_packed(rows):
    """Should a decode forward of ``rows`` rows keep its activations k-octet-major?"""
    if ACT_LAYOUT == "packed":
        return True
    if ACT_LAYOUT == "rows":
        return False
    return rows >= ACT_PACKED_MIN_ROWS


class Act:
    """A k-octet-major activation block: ``t`` is a contiguous (K/8, R, 8) fp16 tensor holding rows 0..M-1 of an (M, K)
    block (R >= M).  Quacks like the (M, K) tensor where the model code looks at it (shape, device, dtype, clone)."""
    __slots__ = ("t", "M", "K")

    def __init__(self, t, M):
        assert t.dim() == 3 and t.shape[2] == 8 and t.is_contiguous() and t.dtype == _HALF and t.shape[1] >= M
        self.t, self.M, self.K = t, M, t.shape[0] * 8

    @classmethod
    def empty(cls, M, K, device, R=None):
        return cls(torch.empty(K // 8, M if R is None else R, 8, dtype=_HALF, device=device), M)

    @classmethod
    def from_rows(cls, x, R=None):
        return cls(pack_act(x, R), x.shape[0])

    @classmethod
    def over(cls, flat, M, K):
        """View a flat fp16 buffer of M * K elements (e.g. the all-reduce staging area) as an M-row block."""
        assert flat.numel() == M * K and flat.is_contiguous()
        return cls(flat.view(K // 8, M, 8), M)

    R = property(lambda self: self.t.shape[1])
    sm = property(lambda self: 8)
    sk = property(lambda self: 8 * self.t.shape[1])
    shape = property(lambda self: (self.M, self.K))
    device = property(lambda self: self.t.device)
    dtype = property(lambda self: self.t.dtype)
    is_cuda = property(lambda self: self.t.is_cuda)

    def data_ptr(self):
        return self.t.data_ptr()

    def numel(self):
        return self.M * self.K

    def rows(self):
        """Row-major (M, K) copy."""
        return unpack_act(self.t, self.M)

    def clone(self):
        return Act(self.t.clone(), self.M)

    def copy_(self, other):
        if isinstance(other, Act):
            assert other.shape == self.shape and other.R == self.R
            self.t.copy_(other.t)
        else:
            self.t[:, :self.M].copy_(other.view(self.M, self.K // 8, 8).permute(1, 0, 2))
        return self


def _lay(x):
    """(pointer, sm, sk) of an activation operand: an Act, a row-major 2-D tensor, or None."""
    if x is None:
        return None, 8, 8
    if isinstance(x, Act):
        return _ptr(x.t), x.sm, x.sk
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == _HALF
    return _ptr(x), x.stride(0), 8


def pack_act(x, R=None):
    """Row-major (M, K) fp16 -> k-octet-major block (K/8, R, 8): element (m, k) at [(k // 8), m, k % 8]; R >= M rows
    (rows M..R-1 are zero).  Strides for the *_act entry points: s_m = 8, s_k = 8 * R."""
    M, K = x.shape
    R = M if R is None else R
    assert K % 8 == 0 and R >= M and x.dtype == _HALF
    out = torch.zeros(K // 8, R, 8, dtype=_HALF, device=x.device)
    out[:, :M] = x.reshape(M, K // 8, 8).permute(1, 0, 2)
    return out


def unpack_act(xp, M):
    """Inverse of pack_act: (K/8, R, 8) -> row-major (M, K)."""
    K8, R, _ = xp.shape
    return xp[:, :M].permute(1, 0, 2).reshape(M, K8 * 8).contiguous()


def embed_rows(embed, ids, packed, out=None):
    """x = embed[ids] for the <= 32 rows of a decode forward: a row-major tensor, or (packed) an Act written by
    tf_embed_rows straight in the k-octet-major form.  ``out``: write into this tensor / Act (static graph buffers)."""
    ids = ids.reshape(-1)
    if not packed:
        if embed.is_cuda and embed.dtype == _HALF and embed.is_contiguous() and ids.numel() <= 32 and ids.dtype == torch.int64 \
                and ids.is_contiguous() and embed.shape[1] % 8 == 0 and (out is None or (out.is_contiguous() and out.dtype == _HALF)):
            # the same kernel with row-major strides (round 5: no torch gather kernel left inside the captured decode forwards)
            n, (V, hid) = ids.numel(), embed.shape
            x = torch.empty(n, hid, dtype=_HALF, device=embed.device) if out is None else out
            assert tuple(x.shape) == (n, hid)
            hip.check(hip.lib().tf_embed_rows(_ptr(embed), _ptr(ids), _ptr(x), hid, 8, n, hid, V, _stream()), "tf_embed_rows")
            return x
        if out is None:
            return embed[ids]
        return out.copy_(embed[ids])
    _dev(embed, ids)
    assert embed.dtype == _HALF and embed.is_contiguous() and ids.dtype == torch.int64 and ids.is_contiguous()
    n, (V, hid) = ids.numel(), embed.shape
    x = Act.empty(n, hid, embed.device) if out is None else out
    assert isinstance(x, Act) and x.shape == (n, hid)
    hip.check(hip.lib().tf_embed_rows(_ptr(embed), _ptr(ids), _ptr(x.t), x.sm, x.sk, n, hid, V, _stream()), "tf_embed_rows")
    return x


def set_tokens(dst, vals, pad, pos=None, pos0=0, slot=None, sk=None, sk_val=0):
    """dst (int64, <= 32 entries) = vals padded with ``pad``; pos (int64) = pos0 + arange; slot / sk (int32 scalars) =
    pos0 / sk_val — one launch, the ids passed as kernel arguments (tf_set_tokens).  ``vals``: python ints."""
    n_dst = 0 if dst is None else dst.numel()
    n_vals = len(vals)
    assert n_vals <= 32 and n_dst <= 32 and (dst is None or (dst.dtype == torch.int64 and dst.is_contiguous()))
    assert pos is None or (pos.dtype == torch.int64 and pos.is_contiguous() and pos.numel() <= 64)
    assert (slot is None or slot.dtype == torch.int32) and (sk is None or sk.dtype == torch.int32)
    _dev(dst, pos, slot, sk)
    host = (ctypes.c_int64 * max(n_vals, 1))(*[int(v) for v in vals])
    hip.check(hip.lib().tf_set_tokens(_ptr(dst), n_dst, host, n_vals, int(pad), _ptr(pos), 0 if pos is None else pos.numel(),
                                      int(pos0), _ptr(slot), _ptr(sk), int(sk_val), _stream()), "tf_set_tokens")


def can_fuse(x, *ws):
    """True when the fused decode kernels apply: HIP tensors, <= 32 rows, every weight packed."""
    return (x.is_cuda and x.dim() == 2 and x.shape[0] <= SKINNY_MAX_ROWS and x.dtype == _HALF and x.stride(1) == 1
            and all(isinstance(w, PackedLinear) and w.parts is not None for w in ws))


def can_fuse_rows(rows, embed, *ws):
    """can_fuse before the embedding rows exist: ``rows`` rows of ``embed`` (device fp16 table) against packed weights."""
    return (embed.is_cuda and embed.dtype == _HALF and rows <= SKINNY_MAX_ROWS
            and all(isinstance(w, PackedLinear) and w.parts is not None for w in ws))


def linear(x, w, out_f32=False, ln=None, eps=0.0, resid=None, out=None, ss_in=None, ss_out=None):
    """y = x . W^T, fp16 with fp32 accumulation — the reference's nn.Linear / F.linear.  <=32 rows against a
    PackedLinear run the hand-written weight-streaming kernel; larger blocks (prefill) go to hipBLASLt.
    Fused forms of the skinny kernel (only valid when ``can_fuse``): ``ln`` = RMSNorm weight applied to x first
    (h = ln * fp16(x * rsqrt(mean(x^2)+eps))), ``resid`` = fp16 residual added to the fp16 result, ``out`` = where
    to write (may be ``resid`` itself); ``ss_out`` (N/16, 32) fp32 receives the per-panel sums of squares of the
    output rows and ``ss_in`` feeds such partials of x to the norm prologue (see ``ss_buffer``).
    x / resid / out may be ``Act`` blocks (k-octet-major); an Act input gives an Act output unless ``out`` says
    otherwise (fp32 logits are always a row-major tensor)."""
    M = x.shape[0]
    if isinstance(w, PackedLinear) and w.wp is not None and M <= SKINNY_MAX_ROWS and x.is_cuda:
        assert x.dtype == _HALF and x.shape[1] == w.K
        if out is None:
            if out_f32 or not isinstance(x, Act):
                out = torch.empty(M, w.N, dtype=torch.float32 if out_f32 else _HALF, device=x.device)
            else:
                out = Act.empty(M, w.N, x.device)
        assert tuple(out.shape) == (M, w.N)
        if resid is not None:
            assert tuple(resid.shape) == tuple(out.shape) and resid.dtype == _HALF and not out_f32
        if ss_in is not None:
            assert ss_in.dtype == torch.float32 and ss_in.shape == (w.K // 16, 32) and ss_in.is_contiguous()
        if ss_out is not None:
            assert ss_out.dtype == torch.float32 and ss_out.shape == (w.N // 16, 32) and ss_out.is_contiguous()
        xp, xsm, xsk = _lay(x)
        rp, rsm, rsk = _lay(resid)
        if out_f32:
            assert out.dtype == torch.float32 and out.stride(1) == 1
            yp, ysm, ysk = _ptr(out), out.stride(0), 8
        else:
            yp, ysm, ysk = _lay(out)
        hip.check(hip.lib().tf_skinny_gemm_act(_ptr(w.wp), xp, xsm, xsk, _ptr(ln), float(eps), _ptr(ss_in), rp, rsm, rsk,
                                               _ptr(ss_out), yp, ysm, ysk, M, w.N, w.K, 1 if out_f32 else 0, _stream()),
                  "tf_skinny_gemm_act")
        return out
    assert ln is None and resid is None and out is None and ss_in is None and ss_out is None, \
        "fused linear needs the skinny kernel (ops.can_fuse)"
    assert not isinstance(x, Act), "k-octet-major activations exist only on the skinny decode path"
    y = F.linear(x, _w(w))
    return y.float() if out_f32 else y


def mlp_act(h, wgu, ln=None, eps=0.0, ss_in=None):
    """fp16(silu(gate(h))) * up(h) for a fused gate|up weight: one kernel for <=32 rows (optionally with the
    RMSNorm of h folded in, ``ln``), GEMM + silu_mul otherwise.  An Act input gives an Act output."""
    M = h.shape[0]
    if isinstance(wgu, PackedLinear) and wgu.parts is not None and wgu.split == 2 and M <= SKINNY_MAX_ROWS \
            and h.is_cuda:
        I = wgu.N // 2
        act = Act.empty(M, I, h.device) if isinstance(h, Act) else torch.empty(M, I, dtype=_HALF, device=h.device)
        hp, hsm, hsk = _lay(h)
        ap, asm, ask = _lay(act)
        if wgu.parts_n8 is not None and ln is not None and M <= N8_MAX_ROWS:        # few-panel shard: 8-row panels
            hip.check(hip.lib().tf_skinny_gemm_swiglu_n8(_ptr(wgu.parts_n8[0]), _ptr(wgu.parts_n8[1]), hp, hsm, hsk, _ptr(ln),
                                                         float(eps), _ptr(ss_in), ap, asm, ask, M, I, wgu.K, _stream()),
                      "tf_skinny_gemm_swiglu_n8")
            return act
        hip.check(hip.lib().tf_skinny_gemm_swiglu_act(_ptr(wgu.parts[0]), _ptr(wgu.parts[1]), hp, hsm, hsk, _ptr(ln),
                                                      float(eps), _ptr(ss_in), ap, asm, ask, M, I, wgu.K, _stream()),
                  "tf_skinny_gemm_swiglu_act")
        return act
    assert ln is None and not isinstance(h, Act), "fused mlp_act needs the skinny kernel (ops.can_fuse)"
    return silu_mul(F.linear(h, _w(wgu)))


def ss_buffer(hidden, device):
    """Hand-off buffer for the residual stream's per-panel sums of squares (hidden/16 panels x 32 rows, fp32)."""
    return torch.empty(hidden // 16, 32, dtype=torch.float32, device=device)


def qkv_rope(x, wqkv, ln, eps, cos, sin, positions, k_layer, v_layer, slot0, H, D, rotate_k=True, slot0_dev=None,
             ss_in=None):
    """One kernel for [RMSNorm ->] fused q|k|v GEMM -> RoPE -> KV append: x (rows, hidden) is the residual stream
    (ln = input_layernorm weight, or None when x is already normalised; a row-major tensor or an Act); q (rows,H,D)
    is returned rotated, the k (rotated unless rotate_k is False) and v rows land in the cache at slot0+i."""
    _dev(ln, cos, sin, positions, k_layer, v_layer, slot0_dev)
    assert isinstance(wqkv, PackedLinear) and wqkv.wp_rope is not None and wqkv.rope == (H, D)
    rows = x.shape[0]
    assert x.is_cuda and x.dtype == _HALF and x.shape[1] == wqkv.K and rows <= SKINNY_MAX_ROWS
    assert positions.dtype == torch.int64 and positions.numel() == rows and positions.is_contiguous()
    assert cos.dtype == _HALF and cos.is_contiguous() and cos.shape[1] == D
    st, sh = _kv(k_layer)
    assert _kv(v_layer) == (st, sh)
    q = torch.empty(rows, H, D, dtype=_HALF, device=x.device)
    xp, xsm, xsk = _lay(x)
    if wqkv.wp_rope_n8 is not None and ln is not None and rows <= N8_MAX_ROWS:      # few-panel shard: 8-row panels
        hip.check(hip.lib().tf_skinny_qkv_rope_n8(_ptr(wqkv.wp_rope_n8), xp, xsm, xsk, _ptr(ln), float(eps), _ptr(ss_in),
                                                  _ptr(cos), _ptr(sin), _ptr(positions), _ptr(q), _ptr(k_layer),
                                                  _ptr(v_layer), st, sh, int(slot0), _ptr(slot0_dev), rows, H, D, wqkv.K,
                                                  1 if rotate_k else 0, _stream()), "tf_skinny_qkv_rope_n8")
        return q
    hip.check(hip.lib().tf_skinny_qkv_rope_act(_ptr(wqkv.wp_rope), xp, xsm, xsk, _ptr(ln), float(eps), _ptr(ss_in),
                                               _ptr(cos), _ptr(sin), _ptr(positions), _ptr(q), _ptr(k_layer),
                                               _ptr(v_layer), st, sh, int(slot0), _ptr(slot0_dev), rows, H, D, wqkv.K,
                                               1 if rotate_k else 0, _stream()), "tf_skinny_qkv_rope_act")
    return q


def rmsnorm(x, w, eps, residual=None, sum_out=None):
    """y = w * fp16((x [+ residual]) * rsqrt(mean(.^2)+eps)); writes the fp16 sum to sum_out if given."""
    _dev(x, w, residual, sum_out)
    assert x.dtype == _HALF and x.is_contiguous() and x.dim() == 2
    y = torch.empty_like(x)
    hip.check(hip.lib().tf_rmsnorm(_ptr(x), _ptr(residual), _ptr(w), _ptr(y), _ptr(sum_out), x.shape[0], x.shape[1],
                                   float(eps), _stream()), "tf_rmsnorm")
    return y


def rope_append(qkv, cos, sin, positions, k_layer, v_layer, slot0, H, D, rotate_k=True, slot0_dev=None):
    """Split fused qkv rows, rotate q (and k), append k/v rows to the cache at slot0+i.  Returns q (rows,H,D)."""
    _dev(qkv, cos, sin, positions, k_layer, v_layer)
    rows = qkv.shape[0]
    assert qkv.dtype == _HALF and qkv.stride(1) == 1 and qkv.shape[1] == 3 * H * D
    assert positions.dtype == torch.int64 and positions.numel() == rows and positions.is_contiguous()
    assert cos.dtype == _HALF and cos.is_contiguous() and cos.shape[1] == D
    st, sh = _kv(k_layer)
    assert _kv(v_layer) == (st, sh)
    q = torch.empty(rows, H, D, dtype=_HALF, device=qkv.device)
    hip.check(hip.lib().tf_rope_append(_ptr(qkv), qkv.stride(0), _ptr(cos), _ptr(sin), _ptr(positions), _ptr(q),
                                       _ptr(k_layer), _ptr(v_layer), st, sh, int(slot0), _ptr(slot0_dev), rows, H, D,
                                       1 if rotate_k else 0, _stream()), "tf_rope_append")
    return q


def silu_mul(gate_up):
    _dev(gate_up)
    rows, two_i = gate_up.shape
    assert gate_up.dtype == _HALF and gate_up.is_contiguous()
    out = torch.empty(rows, two_i // 2, dtype=_HALF, device=gate_up.device)
    hip.check(hip.lib().tf_silu_mul(_ptr(gate_up), _ptr(out), rows, two_i // 2, _stream()), "tf_silu_mul")
    return out


def _workspace(device, floats):
    # per-call allocation from torch's caching allocator: no hipMalloc after warm-up, and inside a
    # hipGraph capture the block comes from the graph's private pool, so replays stay valid
    return torch.empty(int(floats), dtype=torch.float32, device=device)


# When set to a list, eager attn_decode calls append (start_event, end_event, sk, H, D): HIP events on the stream
# the kernel is launched on, used by bench.py for the live roofline figure.  Every ATTN_TIMER_EVERY-th call is
# timed: an event record is a queue packet of its own (~3 us of gap on each side of the kernel), so bracketing all
# 32 launches of a target verify would cost the step ~0.4 ms of the very time being measured.
ATTN_TIMER = None
ATTN_TIMER_EVERY = max(1, int(_os.environ.get("TRIFORCE_ATTN_TIMER_EVERY", "8")))
_attn_calls = 0


_NSPLIT_FORCE = int(_os.environ.get("TRIFORCE_ATTN_NSPLIT", "0"))      # A/B only (tools/tp_shard_bench.py): > 0 overrides the rule


@functools.lru_cache(maxsize=4096)
def _pick_nsplit(H, sk):
    if _NSPLIT_FORCE > 0:
        return max(1, min(_NSPLIT_FORCE, (int(sk) + 15) // 16))
    return hip.lib().tf_attn_decode_pick_nsplit(H, sk)


@functools.lru_cache(maxsize=4096)
def _ws_floats(H, sq, D, nsplit):
    return hip.lib().tf_attn_decode_ws_floats(H, sq, D, nsplit)


# Ticket words of the one-launch attention (tf_attn_decode_fused): per device one zeroed block, one 128-word row per
# stream that has launched attention (calls on one stream are ordered; two streams must not share a row).  Allocated
# at the first call, never freed; every launch leaves its row zero.
ATTN_FUSED_MERGE = _os.environ.get("TRIFORCE_ATTN_FUSED_MERGE", "1") != "0"
if _os.environ.get("TRIFORCE_ATTN_RENDEZVOUS", "0") == "1":     # A/B: > 8 splits on a small grid merged inside the launch
    hip.lib().tf_attn_tune(0, 1)
_TICKET_ROWS, _TICKET_WORDS = 64, 128
_tickets = {}


def _ticket_row(device, stream):
    dev = _tickets.get(device)
    if dev is None:
        dev = _tickets[device] = (torch.zeros(_TICKET_ROWS, _TICKET_WORDS, dtype=torch.int32, device=device), {})
    block, rows = dev
    row = rows.get(stream)
    if row is None:
        if len(rows) >= _TICKET_ROWS:
            return None                               # more streams than rows: the two-launch form is always valid
        row = rows[stream] = len(rows)
    return block[row]


def attn_decode(q, k_layer, v_layer, sk, scale, sk_dev=None, nsplit=None, packed=False):
    """flash_attn_with_kvcache(q, k, v, softmax_scale, causal=True) for sq<=32 rows (bottom-right causal).
    q (sq,H,D); returns (sq, H*D) fp16 — a row-major tensor, or with ``packed`` an Act (what o_proj then reads)."""
    _dev(q, k_layer, v_layer, sk_dev)
    sq, H, D = q.shape
    assert q.dtype == _HALF and q.is_contiguous()
    st, sh = _kv(k_layer)
    assert _kv(v_layer) == (st, sh)
    L = hip.lib()
    if nsplit is None:
        nsplit = _pick_nsplit(H, int(sk))
    ws = _workspace(q.device, _ws_floats(H, sq, D, nsplit))
    out = Act.empty(sq, H * D, q.device) if packed else torch.empty(sq, H * D, dtype=_HALF, device=q.device)
    op, osm, osk = _lay(out)
    timed = False
    if ATTN_TIMER is not None and not torch.cuda.is_current_stream_capturing():
        global _attn_calls
        _attn_calls += 1
        timed = _attn_calls % ATTN_TIMER_EVERY == 0
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    stream = _stream()
    tickets = _ticket_row(q.device, stream.value or 0) if ATTN_FUSED_MERGE and H <= _TICKET_WORDS else None
    hip.check(L.tf_attn_decode_act(_ptr(q), _ptr(k_layer), _ptr(v_layer), op, osm, osk, st, sh, sq, int(sk), _ptr(sk_dev),
                                   H, D, float(scale), nsplit, _ptr(ws), ws.numel(), _ptr(tickets), stream),
              "tf_attn_decode_act")
    if timed:
        ev1.record()
        ATTN_TIMER.append((ev0, ev1, int(sk), H, D))
    return out


def attn_block(q, k_layer, v_layer, sk, scale, nsplit=None, tree_mask=None, mask_row0=0, tree_start=0):
    """Attention of a block of <=128 query rows in one pass over the keys.  tree_mask None: bottom-right causal
    (a prefill chunk).  tree_mask (n_rows, words) int32 bit rows: Sequoia tree attention — keys [0, tree_start)
    visible to all rows, key tree_start+j visible to row i iff bit j of tree_mask[mask_row0+i] is set."""
    _dev(q, k_layer, v_layer, tree_mask)
    sq, H, D = q.shape
    assert q.dtype == _HALF and q.is_contiguous() and sq <= 128
    st, sh = _kv(k_layer)
    assert _kv(v_layer) == (st, sh)
    L = hip.lib()
    if nsplit is None:
        nsplit = L.tf_attn_block_pick_nsplit(H, sq, int(sk))
    ws = _workspace(q.device, L.tf_attn_block_ws_floats(H, D, nsplit))
    out = torch.empty(sq, H * D, dtype=_HALF, device=q.device)
    words = 0
    if tree_mask is not None:
        assert tree_mask.dtype == torch.int32 and tree_mask.dim() == 2 and tree_mask.is_contiguous()
        assert mask_row0 + sq <= tree_mask.shape[0]
        words = tree_mask.shape[1]
    hip.check(L.tf_attn_block(_ptr(q), _ptr(k_layer), _ptr(v_layer), _ptr(out), st, sh, sq, int(sk), H, D, float(scale),
                              nsplit, _ptr(ws), ws.numel(), _ptr(tree_mask), words, int(mask_row0), int(tree_start),
                              _stream()), "tf_attn_block")
    return out


def attn_tree(q, k_layer, v_layer, sk, scale, tree_mask, tree_start, mask_row0=0):
    """Tree attention for any number of query rows (<=128 per pass)."""
    sq = q.shape[0]
    if sq <= 128:
        return attn_block(q, k_layer, v_layer, sk, scale, tree_mask=tree_mask, mask_row0=mask_row0, tree_start=tree_start)
    outs = []
    for r0 in range(0, sq, 128):
        r1 = min(sq, r0 + 128)
        outs.append(attn_block(q[r0:r1].contiguous(), k_layer, v_layer, sk, scale, tree_mask=tree_mask,
                               mask_row0=mask_row0 + r0, tree_start=tree_start))
    return torch.cat(outs, dim=0)


def pack_tree_mask(visible):
    """(rows, T) bool/0-1 tensor -> (rows, ceil(T/32)) int32 bit rows (bit j%32 of word j//32 = column j)."""
    rows, T = visible.shape
    words = (T + 31) // 32
    v = torch.zeros(rows, words * 32, dtype=torch.int64, device=visible.device)
    v[:, :T] = (visible != 0).to(torch.int64)
    w = (v.view(rows, words, 32) << torch.arange(32, device=visible.device, dtype=torch.int64)).sum(dim=-1)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)            # two's-complement into int32
    return w.to(torch.int32).contiguous()


# 0: a prefill chunk's attention as one tf_attn_block launch per 128 rows (A/B; default: tf_attn_prefill, one launch)
ATTN_PREFILL_ONE_LAUNCH = _os.environ.get("TRIFORCE_PREFILL_ONE_LAUNCH", "1") != "0"


def attn_prefill(q, k_layer, v_layer, sk, scale):
    """Causal attention for a prefill block of any length: <=32 rows use the decode kernel, longer blocks are cut
    (up to 4096 rows) go to tf_attn_prefill: the 128-row blocks of the chunk, block j seeing keys
    [0, sk - sq + end_j), run in one launch and share the KV stream through L2."""
    sq = q.shape[0]
    if sq <= 32:
        return attn_decode(q, k_layer, v_layer, sk, scale)
    if sq <= 128:
        return attn_block(q, k_layer, v_layer, sk, scale)
    if not ATTN_PREFILL_ONE_LAUNCH or sq > 4096:
        outs = []
        for r0 in range(0, sq, 128):
            r1 = min(sq, r0 + 128)
            outs.append(attn_block(q[r0:r1].contiguous(), k_layer, v_layer, sk - (sq - r1), scale))
        return torch.cat(outs, dim=0)
    _dev(q, k_layer, v_layer)
    _, H, D = q.shape
    assert q.dtype == _HALF and q.is_contiguous()
    st, sh = _kv(k_layer)
    assert _kv(v_layer) == (st, sh)
    L = hip.lib()
    nsplit = L.tf_attn_prefill_pick_nsplit(H, sq, int(sk))
    ws = _workspace(q.device, L.tf_attn_prefill_ws_floats(H, sq, D, nsplit))
    out = torch.empty(sq, H * D, dtype=_HALF, device=q.device)
    hip.check(L.tf_attn_prefill(_ptr(q), _ptr(k_layer), _ptr(v_layer), _ptr(out), st, sh, sq, int(sk), H, D, float(scale),
                                nsplit, _ptr(ws), ws.numel(), _stream()), "tf_attn_prefill")
    return out


def draft_model_struct(embed, ln1, wqkv, wo, ln2, wgu, wd, norm, lm_head, cos, sin, H, D, eps, scale):
    """TfDraftModel for tf_draft_forward_68m from the draft's weights (PackedLinear lists); None when a weight is not
    packed for the fused kernels (CPU tensors, odd shapes).  The struct holds raw pointers: keep the tensors alive."""
    L = len(wqkv)
    ws = list(wqkv) + list(wo) + list(wgu) + list(wd) + [lm_head]
    if L > hip.TF_DRAFT_MAX_LAYERS or D != 64 or not all(isinstance(w, PackedLinear) and w.parts is not None for w in ws):
        return None
    if any(w.wp_rope is None for w in wqkv) or any(w.split != 2 for w in wgu) or not embed.is_cuda:
        return None
    m = hip.TfDraftModel()
    m.embed = embed.data_ptr()
    for i in range(L):
        m.ln1[i], m.ln2[i] = ln1[i].data_ptr(), ln2[i].data_ptr()
        m.wqkv[i], m.wo[i], m.wdown[i] = wqkv[i].wp_rope.data_ptr(), wo[i].wp.data_ptr(), wd[i].wp.data_ptr()
        m.wgate[i], m.wup[i] = wgu[i].parts[0].data_ptr(), wgu[i].parts[1].data_ptr()
    m.norm, m.lm_head, m.cos, m.sin = norm.data_ptr(), lm_head.wp.data_ptr(), cos.data_ptr(), sin.data_ptr()
    m.layers, m.hidden, m.heads, m.head_dim = L, embed.shape[1], H, D
    m.inter, m.vocab, m.eps, m.scale = wgu[0].N // 2, lm_head.N, float(eps), float(scale)
    return m


def draft_cache_struct(cache, L):
    """TfDraftCache over the per-layer K / V views of a StreamingLLM cache."""
    c = hip.TfDraftCache()
    st = sh = None
    for i in range(L):
        kl, vl = cache.layer_kv(i)
        _dev(kl, vl)
        assert _kv(vl) == _kv(kl) and (st is None or (st, sh) == _kv(kl))
        st, sh = _kv(kl)
        c.k[i], c.v[i] = kl.data_ptr(), vl.data_ptr()
    c.stride_t, c.stride_h = st, sh
    return c


class DraftPersist:
    """Device state of the one-launch draft forward (tf_draft_forward_68m_persist): the control block (arrival counters,
    launch epoch, sticky error word, top-p scratch) and the activation workspace, both owned by ONE draft model — its
    launches are stream-ordered (eager calls and graph replays of one engine).  ``mirror``: a pinned host word that
    receives the error code without a device read (None until ``enable_mirror``)."""

    def __init__(self, model, device):
        L = hip.lib()
        self.device = torch.device(device)
        self.ctl = torch.zeros(int(L.tf_draft_persist_ctl_bytes()), dtype=torch.uint8, device=self.device)
        self.ws = torch.empty(int(L.tf_draft_persist_ws_bytes(ctypes.byref(model))), dtype=torch.uint8, device=self.device)
        self.mirror = None
        assert self.ctl.data_ptr() % 64 == 0 and self.ws.data_ptr() % 256 == 0
        self.enable_mirror()

    def check(self):
        """Raise if a launch on this control block has timed out (its outputs are NaN from then on): one plain host load."""
        if self.mirror is not None and int(self.mirror[0]) != 0:
            raise hip.TriforceHipError(f"one-launch draft forward: wait on edge {int(self.mirror[0]) - 1} timed out "
                                       "(a workgroup never became resident, or an arrival was lost); outputs are NaN until "
                                       "DraftPersist.reset()")

    def enable_mirror(self):
        if self.mirror is None:
            self.mirror = torch.zeros(1, dtype=torch.int32).pin_memory()
            hip.check(hip.lib().tf_draft_persist_reset(_ptr(self.ctl), self.mirror.data_ptr(), 1), "tf_draft_persist_reset")
        return self.mirror

    def error(self):
        """0, or 1 + the index of the edge whose wait timed out (sticky).  Reads the pinned mirror when there is one (no device
        synchronisation), else the control block (blocking)."""
        if self.mirror is not None:
            return int(self.mirror[0])
        return int(hip.lib().tf_draft_persist_error(_ptr(self.ctl)))

    def reset(self):
        hip.check(hip.lib().tf_draft_persist_reset(_ptr(self.ctl), None, 0), "tf_draft_persist_reset")


DRAFT_PERSIST = _os.environ.get("TRIFORCE_DRAFT_PERSIST", "1") != "0"


def draft_persist_supported(model, n, kv_len):
    return hip.lib().tf_draft_persist_supported(ctypes.byref(model), int(n), int(kv_len)) == 0


def draft_forward(model, cache, ids, slot0, kv_len, probs=None, persist=None):
    """tf_draft_forward_68m: ids (n,) int64 -> fp32 logits (n, vocab) [and the top-p probability row of the last token
    when ``probs`` = (temperature, top_p)] in one native call — ONE launch (tf_draft_forward_68m_persist) when ``persist``
    (a DraftPersist) is given and the shape is one it takes, else the 13-launch chain; the two are bit-identical."""
    _dev(ids)
    n = ids.numel()
    assert ids.dtype == torch.int64 and ids.is_contiguous() and 1 <= n <= SKINNY_MAX_ROWS
    L = hip.lib()
    logits = torch.empty(n, model.vocab, dtype=torch.float32, device=ids.device)
    p = torch.empty(model.vocab, dtype=torch.float32, device=ids.device) if probs is not None else None
    T, top_p = probs if probs is not None else (1.0, 1.0)
    if persist is not None and DRAFT_PERSIST and draft_persist_supported(model, n, kv_len):
        hip.check(L.tf_draft_forward_68m_persist(ctypes.byref(model), ctypes.byref(cache), _ptr(ids), n, int(slot0),
                                                 int(kv_len), _ptr(logits), _ptr(p), float(T), float(top_p),
                                                 _ptr(persist.ws), persist.ws.numel(), _ptr(persist.ctl), _stream()),
                  "tf_draft_forward_68m_persist")
        return logits, p
    nbytes = L.tf_draft_forward_ws_bytes(ctypes.byref(model), n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ids.device)
    hip.check(L.tf_draft_forward_68m(ctypes.byref(model), ctypes.byref(cache), _ptr(ids), n, int(slot0), int(kv_len),
                                     _ptr(logits), _ptr(p), float(T), float(top_p), _ptr(ws), nbytes, _stream()),
              "tf_draft_forward_68m")
    return logits, p


def attn_rope_on_read(q, k_layer, v_layer, cos, sin, kv_len, scale):
    """Draft attention: cached keys are un-rotated and rotated on read with positions 0..kv_len-1."""
    _dev(q, k_layer, v_layer, cos, sin)
    sq, H, D = q.shape
    assert q.dtype == _HALF and q.is_contiguous()
    st, sh = _kv(k_layer)
    assert _kv(v_layer) == (st, sh)
    out = torch.empty(sq, H * D, dtype=_HALF, device=q.device)
    hip.check(hip.lib().tf_attn_rope_on_read(_ptr(q), _ptr(k_layer), _ptr(v_layer), _ptr(cos), _ptr(sin), _ptr(out),
                                             st, sh, sq, int(kv_len), H, D, float(scale), _stream()),
              "tf_attn_rope_on_read")
    return out


def retrieval_score(k_layer, q, chunks, chunk):
    """(H, chunks) fp16 un-scaled scores q . mean(K chunk)."""
    _dev(k_layer, q)
    H, T, D = k_layer.shape
    st, sh = _kv(k_layer)
    assert q.shape == (H, D) and q.dtype == _HALF and q.is_contiguous() and chunks * chunk <= T
    scores = torch.empty(H, chunks, dtype=_HALF, device=q.device)
    hip.check(hip.lib().tf_retrieval_score(_ptr(k_layer), st, sh, _ptr(q), _ptr(scores), chunks, chunk, H, D,
                                           _stream()), "tf_retrieval_score")
    return scores


def retrieval_topk(scores, sets):
    """(H, sets) int32: chunk 0 first, then the sets-1 best of [1,C), descending, ties -> lowest chunk."""
    _dev(scores)
    H, C = scores.shape
    assert scores.dtype == _HALF and scores.is_contiguous()
    idx = torch.empty(H, sets, dtype=torch.int32, device=scores.device)
    hip.check(hip.lib().tf_retrieval_topk(_ptr(scores), _ptr(idx), C, sets, H, _stream()), "tf_retrieval_topk")
    return idx


def retrieval_gather(k_src, v_src, idx, k_dst, v_dst, chunk):
    _dev(k_src, v_src, idx, k_dst, v_dst)
    H, _, D = k_src.shape
    sst, ssh = _kv(k_src)
    assert _kv(v_src) == (sst, ssh)
    dst_t, dsh = _kv(k_dst)
    assert _kv(v_dst) == (dst_t, dsh)
    assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.shape[0] == H
    hip.check(hip.lib().tf_retrieval_gather(_ptr(k_src), _ptr(v_src), sst, ssh, _ptr(idx), _ptr(k_dst), _ptr(v_dst),
                                            dst_t, dsh, idx.shape[1], chunk, H, D, _stream()), "tf_retrieval_gather")


def _lhtd(t):
    assert t.dim() == 4 and t.stride(3) == 1 and t.dtype == _HALF, "(L,H,T,D) fp16 expected"
    return t.stride(0), t.stride(2), t.stride(1)     # stride_l, stride_t, stride_h


def kv_copy_rows(src, dst, src_t0, dst_t0, n):
    """dst[l,h,dst_t0+i] = src[l,h,src_t0+i] for all layers/heads; src/dst are (L,H,T,D) views."""
    if n <= 0:
        return
    _dev(src, dst)
    L, H, _, D = src.shape
    assert dst.shape[0] == L and dst.shape[1] == H and dst.shape[3] == D
    if src_t0 < 0 or dst_t0 < 0 or src_t0 + n > src.shape[2] or dst_t0 + n > dst.shape[2]:
        raise IndexError(f"kv_copy_rows: rows [{src_t0}, {src_t0 + n}) of {src.shape[2]} -> [{dst_t0}, {dst_t0 + n}) of "
                         f"{dst.shape[2]} leave the cache (the kernel does not bounds-check)")
    ssl, sst, ssh = _lhtd(src)
    dsl, dst_t, dsh = _lhtd(dst)
    hip.check(hip.lib().tf_kv_copy_rows(_ptr(src), ssl, sst, ssh, _ptr(dst), dsl, dst_t, dsh, int(src_t0), int(dst_t0),
                                        int(n), L, H, D, _stream()), "tf_kv_copy_rows")


def kv_shift_rows(cache, src_t0, dst_t0, n):
    """In-place move of rows [src_t0, src_t0+n) down to [dst_t0, ...) (overlap allowed), all layers/heads."""
    if n <= 0 or src_t0 == dst_t0:
        return
    _dev(cache)
    L, H, T, D = cache.shape
    if src_t0 < 0 or dst_t0 < 0 or src_t0 + n > T or dst_t0 + n > T:
        raise IndexError(f"kv_shift_rows: rows [{src_t0}, {src_t0 + n}) -> [{dst_t0}, {dst_t0 + n}) leave the {T}-row cache")
    sl, st, sh = _lhtd(cache)
    hip.check(hip.lib().tf_kv_shift_rows(_ptr(cache), sl, st, sh, int(src_t0), int(dst_t0), int(n), L, H, D,
                                         _stream()), "tf_kv_shift_rows")


def kv_copy_rows_pair(src_k, src_v, dst_k, dst_v, src_t0, dst_t0, n):
    """kv_copy_rows for the K and the V tensor of one cache in ONE launch (same shapes / strides), else two."""
    if n <= 0:
        return
    if not (src_k.shape == src_v.shape and dst_k.shape == dst_v.shape and _lhtd(src_k) == _lhtd(src_v)
            and _lhtd(dst_k) == _lhtd(dst_v)):
        kv_copy_rows(src_k, dst_k, src_t0, dst_t0, n)
        kv_copy_rows(src_v, dst_v, src_t0, dst_t0, n)
        return
    _dev(src_k, src_v, dst_k, dst_v)
    L, H, _, D = src_k.shape
    assert dst_k.shape[0] == L and dst_k.shape[1] == H and dst_k.shape[3] == D
    if src_t0 < 0 or dst_t0 < 0 or src_t0 + n > src_k.shape[2] or dst_t0 + n > dst_k.shape[2]:
        raise IndexError(f"kv_copy_rows: rows [{src_t0}, {src_t0 + n}) of {src_k.shape[2]} -> [{dst_t0}, {dst_t0 + n}) of "
                         f"{dst_k.shape[2]} leave the cache (the kernel does not bounds-check)")
    ssl, sst, ssh = _lhtd(src_k)
    dsl, dst_t, dsh = _lhtd(dst_k)
    hip.check(hip.lib().tf_kv_copy_rows_pair(_ptr(src_k), _ptr(src_v), ssl, sst, ssh, _ptr(dst_k), _ptr(dst_v), dsl, dst_t,
                                             dsh, int(src_t0), int(dst_t0), int(n), L, H, D, _stream()),
              "tf_kv_copy_rows_pair")


def kv_shift_rows_pair(k_cache, v_cache, src_t0, dst_t0, n):
    """kv_shift_rows for the K and the V tensor of one cache in ONE launch (same shapes / strides), else two."""
    if n <= 0 or src_t0 == dst_t0:
        return
    if not (k_cache.shape == v_cache.shape and _lhtd(k_cache) == _lhtd(v_cache)):
        kv_shift_rows(k_cache, src_t0, dst_t0, n)
        kv_shift_rows(v_cache, src_t0, dst_t0, n)
        return
    _dev(k_cache, v_cache)
    L, H, T, D = k_cache.shape
    if src_t0 < 0 or dst_t0 < 0 or src_t0 + n > T or dst_t0 + n > T:
        raise IndexError(f"kv_shift_rows: rows [{src_t0}, {src_t0 + n}) -> [{dst_t0}, {dst_t0 + n}) leave the {T}-row cache")
    sl, st, sh = _lhtd(k_cache)
    hip.check(hip.lib().tf_kv_shift_rows_pair(_ptr(k_cache), _ptr(v_cache), sl, st, sh, int(src_t0), int(dst_t0), int(n),
                                              L, H, D, _stream()), "tf_kv_shift_rows_pair")


# ---- launch plans (round 5) ---------------------------------------------------------------------------------------------
# The decode loop issues the same few launches over the same buffers every step — the tail copy of the retrieval cache, the
# draft window shift, the token / position set-up — and the host time between a decision record and the next launch is GPU idle
# time (profiles/r05a_gap_analysis_decode_steps_inner_graph.txt: 36-108 us in front of each of them under the profiler).  A plan
# validates its tensors ONCE (what the wrappers above do on every call: device, dtype, shapes, strides, slicing views) and keeps
# the ctypes arguments; a call then checks its row range with integer arithmetic and goes straight into the library.
# TRIFORCE_HOST_PLANS=0: the per-call wrappers.
HOST_PLANS = _os.environ.get("TRIFORCE_HOST_PLANS", "1") != "0"


class KvCopyPairPlan:
    """kv_copy_rows_pair(src_k, src_v, dst_k, dst_v, ...) over fixed tensors."""

    def __init__(self, src_k, src_v, dst_k, dst_v):
        assert src_k.shape == src_v.shape and dst_k.shape == dst_v.shape and _lhtd(src_k) == _lhtd(src_v) \
            and _lhtd(dst_k) == _lhtd(dst_v), "K and V must share shapes and strides"
        _dev(src_k, src_v, dst_k, dst_v)
        L, H, _, D = src_k.shape
        assert dst_k.shape[0] == L and dst_k.shape[1] == H and dst_k.shape[3] == D
        self.keep = (src_k, src_v, dst_k, dst_v)
        self.src_rows, self.dst_rows = src_k.shape[2], dst_k.shape[2]
        self.head = (_ptr(src_k), _ptr(src_v)) + _lhtd(src_k) + (_ptr(dst_k), _ptr(dst_v)) + _lhtd(dst_k)
        self.dims = (L, H, D)
        self.fn = hip.lib().tf_kv_copy_rows_pair

    def __call__(self, src_t0, dst_t0, n):
        if n <= 0:
            return
        if src_t0 < 0 or dst_t0 < 0 or src_t0 + n > self.src_rows or dst_t0 + n > self.dst_rows:
            raise IndexError(f"kv_copy_rows: rows [{src_t0}, {src_t0 + n}) of {self.src_rows} -> [{dst_t0}, {dst_t0 + n}) of "
                             f"{self.dst_rows} leave the cache (the kernel does not bounds-check)")
        hip.check(self.fn(*self.head, int(src_t0), int(dst_t0), int(n), *self.dims, _stream()), "tf_kv_copy_rows_pair")


class KvShiftPairPlan:
    """kv_shift_rows_pair(k_cache, v_cache, ...) over fixed tensors."""

    def __init__(self, k_cache, v_cache):
        assert k_cache.shape == v_cache.shape and _lhtd(k_cache) == _lhtd(v_cache), "K and V must share shapes and strides"
        _dev(k_cache, v_cache)
        L, H, T, D = k_cache.shape
        self.keep, self.rows = (k_cache, v_cache), T
        self.head = (_ptr(k_cache), _ptr(v_cache)) + _lhtd(k_cache)
        self.dims = (L, H, D)
        self.fn = hip.lib().tf_kv_shift_rows_pair

    def __call__(self, src_t0, dst_t0, n):
        if n <= 0 or src_t0 == dst_t0:
            return
        if src_t0 < 0 or dst_t0 < 0 or src_t0 + n > self.rows or dst_t0 + n > self.rows:
            raise IndexError(f"kv_shift_rows: rows [{src_t0}, {src_t0 + n}) -> [{dst_t0}, {dst_t0 + n}) leave the {self.rows}-row cache")
        hip.check(self.fn(*self.head, int(src_t0), int(dst_t0), int(n), *self.dims, _stream()), "tf_kv_shift_rows_pair")


class SetTokensPlan:
    """set_tokens over fixed buffers: dst (its first n_dst entries are written, n_dst <= dst.numel()), pos, slot, sk."""

    def __init__(self, dst, pos=None, slot=None, sk=None):
        assert dst is None or (dst.dtype == torch.int64 and dst.is_contiguous() and dst.numel() <= 32)
        assert pos is None or (pos.dtype == torch.int64 and pos.is_contiguous() and pos.numel() <= 64)
        assert (slot is None or slot.dtype == torch.int32) and (sk is None or sk.dtype == torch.int32)
        _dev(dst, pos, slot, sk)
        self.keep = (dst, pos, slot, sk)
        self.n_dst = 0 if dst is None else dst.numel()
        self.p_dst, self.p_pos, self.n_pos = _ptr(dst), _ptr(pos), 0 if pos is None else pos.numel()
        self.p_slot, self.p_sk = _ptr(slot), _ptr(sk)
        self.fn = hip.lib().tf_set_tokens
        self.arr = ctypes.c_int64 * 32

    def __call__(self, vals, pad, pos0=0, sk_val=0, n_dst=None):
        n_vals = len(vals)
        n_dst = self.n_dst if n_dst is None else n_dst
        assert n_vals <= 32 and n_dst <= self.n_dst
        hip.check(self.fn(self.p_dst, n_dst, self.arr(*vals), n_vals, int(pad), self.p_pos, self.n_pos, int(pos0), self.p_slot,
                          self.p_sk, int(sk_val), _stream()), "tf_set_tokens")


def kv_gather_rows(k_cache, v_cache, offset, idx, max_index=None):
    """Rows offset+idx[j] -> offset+j of every layer/head of the (L,H,T,D) K and V views (idx: device int32,
    strictly increasing) — gather_kv_incremental (reference cache.py:333-343).  max_index: the largest entry of idx as
    the host knows it (the list the device tensor was built from), so that the source rows can be bounds-checked
    without reading the device tensor back."""
    _dev(k_cache, v_cache, idx)
    L, H, T, D = k_cache.shape
    sl, st, sh = _lhtd(k_cache)
    assert _lhtd(v_cache) == (sl, st, sh) and idx.dtype == torch.int32 and idx.is_contiguous()
    if offset < 0 or offset + idx.numel() > T or (max_index is not None and offset + max_index >= T):
        raise IndexError(f"kv_gather_rows: {idx.numel()} rows at offset {offset} (largest index {max_index}) leave the "
                         f"{T}-row cache")
    hip.check(hip.lib().tf_kv_gather_rows(_ptr(k_cache), _ptr(v_cache), sl, st, sh, int(offset), _ptr(idx),
                                          idx.numel(), L, H, D, _stream()), "tf_kv_gather_rows")


def sample_without_replacement(logits, rand, k, temperature):
    """(rows*k,) int64: per row the k token ids with the largest log(rand)/softmax(logits/T), descending —
    `(rand.log() / q).topk(k).indices.flatten()` of the reference's sampling callable, one kernel."""
    _dev(logits, rand)
    rows, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and logits.stride(0) == V
    assert rand.dtype == _HALF and rand.shape == logits.shape and rand.stride(1) == 1 and rand.stride(0) == V
    out = torch.empty(rows * k, dtype=torch.int64, device=logits.device)
    hip.check(hip.lib().tf_sample_without_replacement(_ptr(logits), _ptr(rand), _ptr(out), rows, V, int(k),
                                                      float(temperature), _stream()), "tf_sample_without_replacement")
    return out


TREE_ACCEPT_OUT = 64
TREE_ACCEPT_MAX_PATH = 60       # accepted nodes the record can hold (TREE_MAX_PATH in csrc/sampling.hip)


def tree_accept(p_rows, draft_logits, tokens, succ_off, succ, uniforms, temperature, out):
    """Sequoia tree walk on the device: out[64] int64 <- (len(accept_list), next_token, terminal,
    uniforms_consumed, accept_list...); see include/triforce_hip.h."""
    _dev(p_rows, draft_logits, tokens, succ_off, succ, uniforms, out)
    N, V = p_rows.shape
    assert p_rows.dtype == torch.float32 and p_rows.is_contiguous() and draft_logits.shape == p_rows.shape
    assert draft_logits.dtype == torch.float32 and draft_logits.is_contiguous()
    assert tokens.dtype == torch.int64 and tokens.numel() == N and succ_off.dtype == torch.int32
    assert succ_off.numel() == N + 1 and succ.dtype == torch.int32 and uniforms.dtype == torch.float32
    assert out.dtype == torch.int64 and out.numel() >= TREE_ACCEPT_OUT
    hip.check(hip.lib().tf_tree_accept(_ptr(p_rows), _ptr(draft_logits), _ptr(tokens), _ptr(succ_off), _ptr(succ),
                                       _ptr(uniforms), V, float(temperature), _ptr(out), _stream()), "tf_tree_accept")


TOPP_MAX_VOCAB = 32768


TOPP_MULTI = _os.environ.get("TRIFORCE_TOPP_MULTI", "1") != "0"
_topp_multi_state = {}


def _topp_multi(device):
    """(control block, workspace) of tf_topp_probs_multi for this device, allocated at the first call OUTSIDE a capture (a
    graph's warm-up passes always come first); None while capturing without one.  One per device: the launches of a process are
    stream-ordered or synchronised against each other (warm-up stream -> capture -> replays); a caller that runs top-p on two
    streams at once sets TRIFORCE_TOPP_MULTI=0."""
    key = torch.device(device).index or 0
    st = _topp_multi_state.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        L = hip.lib()
        st = (torch.zeros(int(L.tf_topp_multi_ctl_bytes()), dtype=torch.uint8, device=device),
              torch.empty(int(L.tf_topp_multi_ws_bytes(32, 32768)), dtype=torch.uint8, device=device))
        _topp_multi_state[key] = st
    return st


def topp_probs(logits, temperature, top_p, panel_max=None):
    """softmax(top_p_filter(logits / temperature)) for (rows, V) fp32 logits, V <= 32768 — one fused kernel: every row over 16
    workgroups with in-launch hand-offs (tf_topp_probs_multi) where the shape allows, else one workgroup per row
    (tf_topp_probs); bit-identical.  ``panel_max``: the per-panel row maxima the lm_head GEMM left (ops.linear(...,
    out_f32=True, ss_out=...)): the multi-workgroup form then skips its first hand-off."""
    _dev(logits)
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.is_contiguous()
    probs = torch.empty_like(logits)
    rows, V = logits.shape
    # (<= 16 rows: from 17 rows the launch has 8 slices of 4 000 entries per row — 36.8 -> 30.2 us on model-like rows, but
    #  52.6 -> 81.5 on near-flat ones: profiles/r06_topp_one_workgroup_vs_multi.jsonl; those calls keep one workgroup per row)
    if TOPP_MULTI and rows <= 16 and V % 4 == 0 and 64 <= V <= 32768 and logits.data_ptr() % 16 == 0:
        st = _topp_multi(logits.device)
        if st is not None:
            rc = hip.lib().tf_topp_probs_multi(_ptr(logits), _ptr(panel_max), _ptr(probs), rows, V, float(temperature),
                                               float(top_p), _ptr(st[0]), _ptr(st[1]), st[1].numel(), _stream())
            if rc == 0:
                return probs
            if rc != -34:                                        # TF_ERANGE: more workgroups than CUs — the one-per-row kernel
                hip.check(rc, "tf_topp_probs_multi")
    hip.check(hip.lib().tf_topp_probs(_ptr(logits), _ptr(probs), logits.shape[0], logits.shape[1], float(temperature),
                                      float(top_p), _stream()), "tf_topp_probs")
    return probs


def sample_inverse_cdf(probs, u, token_out):
    """token_out[0] <- first index with inclusive cumsum(probs) > u[0]*sum(probs).  All device tensors."""
    _dev(probs, u)
    _dev_or_pinned(token_out)
    assert probs.dtype == torch.float32 and probs.is_contiguous() and probs.dim() == 1
    assert u.dtype == torch.float32 and token_out.dtype == torch.int64
    hip.check(hip.lib().tf_sample_inverse_cdf(_ptr(probs), _ptr(u), _ptr(token_out), probs.numel(), _stream()),
              "tf_sample_inverse_cdf")


def accept_chain(p, q, tokens, uniforms, g2, inclusive, eos_token_id, out):
    """out[4] int64 <- (count, next_token, reason, uniforms_consumed); see include/triforce_hip.h."""
    _dev(p, q, tokens, uniforms)
    _dev_or_pinned(out)
    V = p.shape[-1]
    assert p.dtype == torch.float32 and p.is_contiguous() and p.shape[0] >= g2 + 1
    assert q.dtype == torch.float32 and q.is_contiguous() and q.shape[0] >= g2 and q.shape[-1] == V
    assert tokens.dtype == torch.int64 and tokens.numel() >= g2 and uniforms.numel() >= g2 + 1
    assert out.dtype == torch.int64 and out.numel() >= 4
    hip.check(hip.lib().tf_accept_chain(_ptr(p), _ptr(q), _ptr(tokens), _ptr(uniforms), int(g2), V,
                                        1 if inclusive else 0, int(eos_token_id), _ptr(out), _stream()),
              "tf_accept_chain")


def middle_accept(p, q_d, tokens, uniforms, n, gamma, out):
    """One Middle_Spec step on device: out[3] int64 <- (accepted, follow_up_token, drafted_token)."""
    _dev(p, q_d, tokens, uniforms)
    _dev_or_pinned(out)
    V = p.shape[-1]
    assert p.dtype == torch.float32 and p.is_contiguous() and q_d.dtype == torch.float32 and q_d.numel() == V
    assert tokens.dtype == torch.int64 and tokens.numel() >= gamma + 1 and uniforms.numel() >= 2
    hip.check(hip.lib().tf_middle_accept(_ptr(p), _ptr(q_d), _ptr(tokens), _ptr(uniforms), int(n), int(gamma), V,
                                         _ptr(out), _stream()), "tf_middle_accept")


def mid_record_tokens(rec, tokens, n):
    """tokens[n + 1] (, tokens[n + 2]) <- what the middle_accept record ``rec`` (accepted, follow-up, drafted) implies."""
    _dev(rec, tokens)
    assert rec.dtype == torch.int64 and tokens.dtype == torch.int64 and tokens.is_contiguous() and rec.numel() >= 3
    hip.check(hip.lib().tf_mid_record_tokens(_ptr(rec), _ptr(tokens), tokens.numel(), int(n), _stream()), "tf_mid_record_tokens")


# ---- the same three kernels with their uniforms behind a device cursor: u_k = ubuf[cursor[0] + k] (capturable) -------------
def sample_inverse_cdf_cur(probs, ubuf, cursor, off, token_out):
    """token_out[0] <- sample(probs) with u = ubuf[cursor[0] + off]; the cursor is left alone."""
    _dev(probs, ubuf, cursor)
    _dev_or_pinned(token_out)
    assert probs.dtype == torch.float32 and probs.is_contiguous() and probs.dim() == 1
    assert ubuf.dtype == torch.float32 and cursor.dtype == torch.int64 and token_out.dtype == torch.int64
    hip.check(hip.lib().tf_sample_inverse_cdf_cur(_ptr(probs), _ptr(ubuf), _ptr(cursor), int(off), _ptr(token_out),
                                                  probs.numel(), _stream()), "tf_sample_inverse_cdf_cur")


def middle_accept_cur(p, q_d, tokens, ubuf, cursor, n, gamma, out):
    """middle_accept over ubuf[cursor + 1], ubuf[cursor + 2]; cursor += 3; out[4] <- (accepted, follow-up, drafted, cursor
    value the decision started from)."""
    _dev(p, q_d, tokens, ubuf, cursor)
    _dev_or_pinned(out)
    V = p.shape[-1]
    assert p.dtype == torch.float32 and p.is_contiguous() and q_d.dtype == torch.float32 and q_d.numel() == V
    assert tokens.dtype == torch.int64 and tokens.numel() >= gamma + 1 and out.numel() >= 4
    assert ubuf.dtype == torch.float32 and cursor.dtype == torch.int64
    hip.check(hip.lib().tf_middle_accept_cur(_ptr(p), _ptr(q_d), _ptr(tokens), tokens.numel(), _ptr(ubuf), _ptr(cursor), int(n),
                                             int(gamma), V, _ptr(out), _stream()), "tf_middle_accept_cur")


def accept_chain_cur(p, q, tokens, ubuf, cursor, g2, inclusive, eos_token_id, out):
    """accept_chain over ubuf[cursor ...]; cursor += the numbers consumed (out[3])."""
    _dev(p, q, tokens, ubuf, cursor)
    _dev_or_pinned(out)
    V = p.shape[-1]
    assert p.dtype == torch.float32 and p.is_contiguous() and p.shape[0] >= g2 + 1
    assert q.dtype == torch.float32 and q.is_contiguous() and q.shape[0] >= g2 and q.shape[-1] == V
    assert tokens.dtype == torch.int64 and tokens.numel() >= g2 and out.dtype == torch.int64 and out.numel() >= 4
    assert ubuf.dtype == torch.float32 and cursor.dtype == torch.int64
    hip.check(hip.lib().tf_accept_chain_cur(_ptr(p), _ptr(q), _ptr(tokens), _ptr(ubuf), _ptr(cursor), int(g2), V,
                                            1 if inclusive else 0, int(eos_token_id), _ptr(out), _stream()),
              "tf_accept_chain_cur")


def accept_chain_step(p, q, tok_buf, ubuf, cursor, g2, inclusive, eos_token_id, pad, s_src, sets, out):
    """accept_chain_cur over tok_buf[1:] that also writes the pass tokens into tok_buf and, for each (pos, slot, sk, qlen) in
    ``sets`` (<= 2), the next target verify's positions / append slot / key count (include/triforce_hip.h)."""
    _dev(p, q, tok_buf, ubuf, cursor, s_src)
    _dev_or_pinned(out)
    V = p.shape[-1]
    assert p.dtype == torch.float32 and p.is_contiguous() and p.shape[0] >= g2 + 1
    assert q.dtype == torch.float32 and q.is_contiguous() and q.shape[0] >= g2 and q.shape[-1] == V
    assert tok_buf.dtype == torch.int64 and tok_buf.is_contiguous() and tok_buf.numel() >= g2 + 2
    assert out.dtype == torch.int64 and out.numel() >= 4 and s_src.dtype == torch.int32 and len(sets) <= 2
    flat = []
    for pos, slot, sk, qlen in list(sets) + [(None, None, None, 0)] * (2 - len(sets)):
        assert pos is None or (pos.dtype == torch.int64 and pos.is_contiguous() and slot.dtype == torch.int32 and sk.dtype == torch.int32)
        flat += [_ptr(pos), 0 if pos is None else pos.numel(), _ptr(slot), _ptr(sk), int(qlen)]
    hip.check(hip.lib().tf_accept_chain_step(_ptr(p), _ptr(q), _ptr(tok_buf), tok_buf.numel(), _ptr(ubuf), _ptr(cursor), int(g2), V,
                                             1 if inclusive else 0, int(eos_token_id), int(pad), _ptr(s_src), len(sets), *flat,
                                             _ptr(out), _stream()), "tf_accept_chain_step")


def kv_h2d_async(dst_dev, src_host, n_tokens, stream):
    """Pinned host (H,T,D) -> device (H,T',D): tokens [0, n_tokens) of every head, on `stream` (a torch Stream)."""
    _dev(dst_dev)
    H, _, D = dst_dev.shape
    assert src_host.shape[0] == H and src_host.shape[2] == D and src_host.dtype == _HALF and not src_host.is_cuda
    hip.check(hip.lib().tf_kv_h2d_async(_ptr(dst_dev), dst_dev.stride(0), _ptr(src_host), src_host.stride(0),
                                        int(n_tokens) * D, H, ctypes.c_void_p(stream.cuda_stream)), "tf_kv_h2d_async")


def kv_d2h_async(dst_host, src_dev, t0, n_tokens, stream):
    """Device (H,T',D) tokens [t0, t0+n) -> pinned host (H,T,D) same token range, on `stream`."""
    _dev(src_dev)
    H, _, D = src_dev.shape
    assert dst_host.shape[0] == H and dst_host.shape[2] == D and dst_host.dtype == _HALF and not dst_host.is_cuda
    dst = ctypes.c_void_p(dst_host.data_ptr() + int(t0) * D * 2)
    src = ctypes.c_void_p(src_dev.data_ptr() + int(t0) * D * 2)
    hip.check(hip.lib().tf_kv_d2h_async(dst, dst_host.stride(0), src, src_dev.stride(0), int(n_tokens) * D, H,
                                        ctypes.c_void_p(stream.cuda_stream)), "tf_kv_d2h_async")
