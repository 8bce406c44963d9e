"""CPU stand-ins for triforce_amd.ops built on the oracle (TEST-ONLY; installed by the ``cpu_ops``
fixture).  They take the product's head-major (H,T,D) layer views and call oracle/ref_ops.py."""
import torch

from oracle import ref_ops as R

PATCHED = ["rmsnorm", "rope_append", "silu_mul", "attn_decode", "attn_prefill", "attn_rope_on_read", "retrieval_score",
           "retrieval_topk", "retrieval_gather", "kv_copy_rows", "kv_shift_rows", "sample_inverse_cdf",
           "accept_chain", "middle_accept", "attn_block", "attn_tree", "kv_gather_rows", "tree_accept", "kv_copy_rows_pair",
           "kv_shift_rows_pair", "set_tokens"]


def rmsnorm(x, w, eps, residual=None, sum_out=None):
    if residual is not None:
        s = x + residual
        if sum_out is not None:
            sum_out.copy_(s)
        x = s
    return R.rms_norm(x, w, eps)


def rope_append(qkv, cos, sin, positions, k_layer, v_layer, slot0, H, D, rotate_k=True, slot0_dev=None):
    rows = qkv.shape[0]
    q = qkv[:, :H * D].reshape(rows, H, D)
    k = qkv[:, H * D:2 * H * D].reshape(rows, H, D)
    v = qkv[:, 2 * H * D:].reshape(rows, H, D)
    qr = R.apply_rope(q, cos, sin, positions)
    kr = R.apply_rope(k, cos, sin, positions) if rotate_k else k
    k_layer[:, slot0:slot0 + rows] = kr.permute(1, 0, 2)
    v_layer[:, slot0:slot0 + rows] = v.permute(1, 0, 2)
    return qr.contiguous()


def silu_mul(gate_up):
    I = gate_up.shape[1] // 2
    return R.silu_mul(gate_up[:, :I], gate_up[:, I:])


def attn_decode(q, k_layer, v_layer, sk, scale, sk_dev=None, nsplit=None, packed=False):
    assert not packed, "k-octet-major activations exist only on the device path"
    sq, H, D = q.shape
    k = k_layer[:, :sk].permute(1, 0, 2)
    v = v_layer[:, :sk].permute(1, 0, 2)
    return R.attn_kvcache(q, k, v, scale).reshape(sq, H * D)


def attn_prefill(q, k_layer, v_layer, sk, scale):
    # one shot (the product cuts a block into <=32-row slabs; same math, different fp32 summation order)
    return attn_decode(q, k_layer, v_layer, sk, scale)


def attn_rope_on_read(q, k_layer, v_layer, cos, sin, kv_len, scale):
    sq, H, D = q.shape
    k = R.apply_rope(k_layer[:, :kv_len].permute(1, 0, 2), cos, sin, torch.arange(kv_len))
    v = v_layer[:, :kv_len].permute(1, 0, 2)
    return R.attn_kvcache(q, k, v, scale).reshape(sq, H * D)


def retrieval_score(k_layer, q, chunks, chunk):
    return R.retrieval_scores(k_layer.permute(1, 0, 2), q, chunks * chunk, chunk)


def retrieval_topk(scores, sets):
    return R.retrieval_topk(scores, sets).to(torch.int32)


def retrieval_gather(k_src, v_src, idx, k_dst, v_dst, chunk):
    n = idx.shape[1] * chunk
    k_dst[:, :n] = R.retrieval_gather(k_src.permute(1, 0, 2), idx.long(), chunk).permute(1, 0, 2)
    v_dst[:, :n] = R.retrieval_gather(v_src.permute(1, 0, 2), idx.long(), chunk).permute(1, 0, 2)


def kv_copy_rows(src, dst, src_t0, dst_t0, n):
    if n > 0:
        dst[:, :, dst_t0:dst_t0 + n] = src[:, :, src_t0:src_t0 + n]


def kv_shift_rows(cache, src_t0, dst_t0, n):
    if n > 0 and src_t0 != dst_t0:
        cache[:, :, dst_t0:dst_t0 + n] = cache[:, :, src_t0:src_t0 + n].clone()


def kv_copy_rows_pair(src_k, src_v, dst_k, dst_v, src_t0, dst_t0, n):
    kv_copy_rows(src_k, dst_k, src_t0, dst_t0, n)
    kv_copy_rows(src_v, dst_v, src_t0, dst_t0, n)


def kv_shift_rows_pair(k_cache, v_cache, src_t0, dst_t0, n):
    kv_shift_rows(k_cache, src_t0, dst_t0, n)
    kv_shift_rows(v_cache, src_t0, dst_t0, n)


def set_tokens(dst, vals, pad, pos=None, pos0=0, slot=None, sk=None, sk_val=0):
    if dst is not None:
        flat = dst.view(-1)
        flat.fill_(pad)
        if len(vals):
            flat[:len(vals)] = torch.tensor([int(v) for v in vals], dtype=flat.dtype)
    if pos is not None:
        pos.view(-1).copy_(torch.arange(pos.numel(), dtype=pos.dtype) + int(pos0))
    if slot is not None:
        slot.fill_(int(pos0))
    if sk is not None:
        sk.fill_(int(sk_val))


def sample_inverse_cdf(probs, u, token_out):
    token_out[0] = R.sample_inverse_cdf(probs, float(u[0]))


def accept_chain(p, q, tokens, uniforms, g2, inclusive, eos_token_id, out):
    c, nxt, reason, consumed = R.accept_and_correct(p[:g2 + 1], q[:g2], tokens[:g2].tolist(),
                                                    uniforms[:g2 + 1].tolist(), inclusive, eos_token_id)
    out[:4] = torch.tensor([c, nxt, reason, consumed])


def middle_accept(p, q_d, tokens, uniforms, n, gamma, out):
    d = int(tokens[n + 1])
    acc, b = R.middle_accept(p, q_d, d, n, uniforms[:2].tolist())
    out[:3] = torch.tensor([acc, b, d])
    if n + 1 + acc <= gamma:
        tokens[n + 1 + acc] = b


# ---- Sequoia tree path ----------------------------------------------------------------------
def _unpack_bits(bits, T):
    w = bits.to(torch.int64) & 0xFFFFFFFF
    cols = torch.arange(T)
    return ((w[:, cols // 32] >> (cols % 32)) & 1).bool()


def attn_block(q, k_layer, v_layer, sk, scale, nsplit=None, tree_mask=None, mask_row0=0, tree_start=0):
    if tree_mask is None:
        return attn_decode(q, k_layer, v_layer, sk, scale)
    from oracle import ref_tree as RT
    sq, H, D = q.shape
    assert abs(scale - D ** -0.5) < 1e-9, "tree attention uses SDPA's default 1/sqrt(D) scale"
    vis = _unpack_bits(tree_mask[mask_row0:mask_row0 + sq], sk - tree_start)
    add = torch.zeros(sq, sk, dtype=torch.float16)
    add[:, tree_start:] = torch.where(vis, 0.0, torch.finfo(torch.float16).min).to(torch.float16)
    k = k_layer[:, :sk].permute(1, 0, 2)
    v = v_layer[:, :sk].permute(1, 0, 2)
    return RT.attn_sdpa(q, k, v, add).reshape(sq, H * D)


def attn_tree(q, k_layer, v_layer, sk, scale, tree_mask, tree_start, mask_row0=0):
    return attn_block(q, k_layer, v_layer, sk, scale, tree_mask=tree_mask, mask_row0=mask_row0, tree_start=tree_start)


def kv_gather_rows(k_cache, v_cache, offset, idx, max_index=None):
    src = [offset + int(i) for i in idx.tolist()]
    for t in (k_cache, v_cache):
        t[:, :, offset:offset + len(src)] = t[:, :, src].clone()


def tree_accept(p_rows, draft_logits, tokens, succ_off, succ, uniforms, temperature, out):
    from oracle import ref_model as M
    from oracle import ref_tree as RT
    so, sc = succ_off.tolist(), succ.tolist()
    successors = [sc[so[i]:so[i + 1]] for i in range(len(so) - 1)]
    rng = M.InjectedRng(uniforms.tolist())
    acc_list, nxt, terminal, _ = RT.accept_walk(p_rows, draft_logits, tokens, successors, temperature, rng)
    out.zero_()
    out[0], out[1], out[2], out[3] = len(acc_list), (0 if nxt is None else nxt), int(terminal), rng.i
    out[4:4 + len(acc_list)] = torch.tensor(acc_list)
