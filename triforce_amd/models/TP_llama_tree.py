"""Tensor-parallel engine of the Sequoia tree path — host-side mirror of the reference's models/TP_llama_tree.py
(DistributedLlama :27-425: same constructor keywords incl. ``tree_size``, same methods reset / prefill /
build_retrieval_cache / inference(attention_mask=) / retrieval_tree_inference).

What differs from models/TP_llama.py (exactly as in the reference):
  * the retrieval cache holds one KV row per TREE NODE behind the budget slots
    (DistributedRetrievalCache_Seqouia, cache.py:385-483) and the full cache has ``tree_size`` extra rows;
  * ``retrieval_tree_inference`` runs the target weights over the retrieval cache for the nodes of one tree
    level (rows written at ``storage_ids``, attention restricted by the tree mask) — tensor_op.py:230-272;
  * ``inference(attention_mask=...)`` verifies the whole tree against the full KV cache in one forward
    (TP_llama_tree.py:179-219 -> tensor_op.py:169-172).

The reference runs both through ``F.scaled_dot_product_attention`` with a dense additive mask
(q x (S + tree) fp16, 133 MB per verify at S = 130K); here the mask is 512-bit rows read only for the last
``tree_size`` keys by the block-attention kernel (tf_attn_block), and the prefix streams unmasked.
"""
import torch

from .. import ops
from .TP_llama import DistributedLlama as _DistributedLlama
from .TP_llama import TreeMask, distributed_init  # noqa: F401  (re-exported like the reference module)


class DistributedLlama(_DistributedLlama):
    def __init__(self, model_name_or_path: str, dtype=torch.float16, kv_offload=False, on_chip_layers=32, local_rank=0,
                 world_size=1, prefill=32768, bsz=1, gen_len=256, retrieval_budget=4096, retrieval_chunk_size=8, gamma=6,
                 temperature=0.6, top_p=0.9, tree_size=128, ssl=0, flash_attn=True, config=None, device=None) -> None:
        assert ssl == 0, "self-speculation layers (ssl) are unused by the entry point (offloading_seqouia.py:117)"
        super().__init__(model_name_or_path, dtype=dtype, kv_offload=kv_offload, on_chip_layers=on_chip_layers,
                         local_rank=local_rank, world_size=world_size, prefill=prefill, bsz=bsz, gen_len=gen_len,
                         retrieval_budget=retrieval_budget, retrieval_chunk_size=retrieval_chunk_size, gamma=gamma,
                         temperature=temperature, top_p=top_p, ssl=ssl, flash_attn=flash_attn, config=config,
                         device=device, tree_size=tree_size)

    @torch.inference_mode()
    def retrieval_tree_inference(self, input_ids, storage_ids, position_ids, attention_mask):
        """Target weights over the retrieval cache for the n nodes of one tree level (TP_llama_tree.py:406-425).
        storage_ids: the n consecutive retrieval-cache slots of those nodes (a range, a list or a tensor);
        attention_mask: TreeMask or the reference's dense (1,1,n,budget+tree) additive mask.  Returns fp32 logits
        (1, n, V)."""
        rc = self.retrieval_cache
        q_len = input_ids.shape[1]
        ids = storage_ids if isinstance(storage_ids, (range, list, tuple)) else storage_ids.tolist()
        slot0 = int(ids[0])
        assert len(ids) == q_len and int(ids[-1]) == slot0 + q_len - 1, "storage_ids must be consecutive slots"
        assert rc.max_budget <= slot0 and slot0 + q_len <= rc.real_budget
        tree = self._tree_mask(attention_mask, rc.max_budget, q_len)
        pos = position_ids.reshape(-1).contiguous()
        x = self.embed_tokens[input_ids.reshape(-1)]
        d = None
        for idx in range(self.num_layers):
            kl, vl = rc.layer_kv(idx)
            d = self._layer(idx, x, d, pos, kl, vl, slot0, rc.real_budget, q_len, tree=tree)
        return self._finish(x, d)


__all__ = ["DistributedLlama", "TreeMask", "distributed_init", "ops"]
