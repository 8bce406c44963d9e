import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
USE_DIST = os.environ.get("DBG_DIST", "0") == "1"
if USE_DIST:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
    dist.init_process_group("nccl", rank=0, world_size=1)
from triforce_amd.models import zoo
from triforce_amd.models.TP_llama_tree import DistributedLlama
from triforce_amd.utils.SpecTree_TP import SpecTree
from triforce_amd.utils.tree import load_grow_map
P, B, G = int(os.environ.get("DBG_P", 4096)), int(os.environ.get("DBG_B", 1024)), int(os.environ.get("DBG_G", 32))
dev = torch.device("cuda:0")
tcfg = zoo.config("llama-7B-128K")
gm = load_grow_map("512")
llm = DistributedLlama("random:1", config=tcfg, local_rank=0, world_size=1, prefill=P, gen_len=G, retrieval_budget=B,
                       kv_offload=True, on_chip_layers=32, tree_size=gm["size"], device=dev)
llm.init_parameters("random:1")
st = SpecTree(engine=llm, grow_map=gm, vocab_size=tcfg.vocab_size)
if "DBG_LEVELS" in os.environ:
    st._debug_max_levels = int(os.environ["DBG_LEVELS"])
if os.environ.get("DBG_NOSAMPLE") == "1":
    st.sampling_callables = {i: (lambda lg, rnd, k=max(st.branches[i]): (torch.arange(lg.shape[0] * k, device=lg.device) % 1000) + 5) for i in range(st.draft_step - 1)}
def sync(tag):
    torch.cuda.synchronize(); print("ok", tag, flush=True)
LATE = os.environ.get("DBG_LATE", "0") == "1"
if os.environ.get("DBG_GRAPH", "0") == "1" and not LATE:
    st.capture_grow_graph(); sync("capture")
ids = torch.randint(3, 32000, (P,), generator=torch.Generator().manual_seed(0)).to(dev)
nt = st.prefill(ids); sync("prefill")
if os.environ.get("DBG_GRAPH", "0") == "1" and LATE:
    st.capture_grow_graph(); sync("late capture")
n = 0
while n < G and n < int(os.environ.get("DBG_STEPS", 99)):
    st.construct_grow_map(nt); sync(f"grow n={n}")
    nt, acc, toks = st.verify(); sync(f"verify n={n} acc={acc} S={llm.kv_cache.seq_len}")
    if nt is None: break
    nt = nt.unsqueeze(0); n += acc
print("DONE", n)
