"""triforce_amd — MI355X-native (gfx950) implementation of the TriForce hierarchical
draft / retrieval-verify / full-verify decode path.

Host side (this package) mirrors the reference's Python interface for that path
(models.cache, models.modeling_llama[_68m], models.TP_llama, utils.graph_infer, utils.decoding,
utils.sampling); the compute is hand-written HIP behind the C ABI of include/triforce_hip.h.
"""
__version__ = "0.1.0"
