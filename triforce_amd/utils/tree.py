"""Static speculation trees for the Sequoia path — the data the reference loads from ``tree/<size>.pt``
(test/offloading_seqouia.py:73-79) and builds offline with tree/tree_search.py.

A grow map is a dict with the reference's schema:
  roots       list over tree levels of the node ids on that level (level 0 = [0])
  branches    list over levels of the child count of every node on that level
  Successors  list over nodes of their child ids (ids are assigned level by level, parents in order)
  mask        (N, N) int64, mask[i, j] = 1 iff j is i or an ancestor of i
  depth       (N,) int64 depth of every node
  size        N

``grow_map_from_branches`` rebuilds all of it from ``branches`` alone (tests pin it against the reference's
512-node fixture); ``search_tree`` is our own dynamic program for the same objective as tree_search.py
(expected accepted tokens of a tree with m nodes / depth <= l given the rank-wise acceptance vector),
vectorised over the split point instead of the reference's Python triple loop.
"""
import json
import os

import numpy as np
import torch

# Rank-wise acceptance rates measured by the reference authors (tree/acceptance-rate-vector.pt: p[b] = chance
# that the b-th child is the accepted one); used only to pick a default tree shape when no file is given.
DEFAULT_ACCEPTANCE = [0.0, 0.9169, 0.04625, 0.01472, 0.006938, 0.004903, 0.002582, 0.002362, 0.001582,
                      0.000806, 0.000488, 0.000479]


def grow_map_from_branches(branches):
    """branches[level][j] = number of children of the j-th node of that level -> full grow map."""
    branches = [list(map(int, b)) for b in branches]
    roots, successors, depth, parents = [[0]], [[]], [0], [-1]
    n = 1
    for lvl, blist in enumerate(branches):
        level_nodes = roots[lvl]
        assert len(blist) == len(level_nodes), f"level {lvl}: {len(blist)} branch counts for {len(level_nodes)} nodes"
        nxt = []
        for node, b in zip(level_nodes, blist):
            kids = list(range(n, n + b))
            successors[node].extend(kids)
            successors.extend([[] for _ in range(b)])
            parents.extend([node] * b)
            depth.extend([depth[node] + 1] * b)
            nxt.extend(kids)
            n += b
        if not nxt:
            break
        roots.append(nxt)
    # the reference's lists end with a level whose branch counts are all zero (tree_search.py:96-118)
    while len(branches) > len(roots):
        branches = branches[:-1]
    if len(branches) < len(roots):
        branches = branches + [[0] * len(roots[-1])]
    mask = torch.zeros(n, n, dtype=torch.int64)
    for i in range(n):
        if parents[i] >= 0:
            mask[i] = mask[parents[i]]
        mask[i, i] = 1
    return {"roots": roots, "branches": branches, "Successors": successors, "mask": mask,
            "depth": torch.tensor(depth, dtype=torch.int64), "size": n}


def successors_csr(successors, device=None):
    """Successors lists -> (offsets int32 [N+1], children int32 [E]) for the device-side tree walk."""
    off = [0]
    flat = []
    for kids in successors:
        flat.extend(int(k) for k in kids)
        off.append(len(flat))
    if not flat:
        flat = [0]
    return (torch.tensor(off, dtype=torch.int32, device=device), torch.tensor(flat, dtype=torch.int32, device=device))


def search_tree(acceptance, max_budget, max_depth):
    """Dynamic program over (nodes m, depth l, root fan-out b):
         G[m][l][1] = 1 + p1 * F[m-1][l-1]
         G[m][l][b] = max_y  G[y][l][b-1] + p_b * F[m-y][l-1]          F[m][l] = max_b G[m][l][b]
    F[m][l] = expected number of accepted tokens (root included) of the best tree with exactly m nodes and depth
    <= l.  Returns (F, back) where back lets ``expand_tree`` rebuild the branch lists."""
    p = np.asarray(acceptance, dtype=np.float64)
    B = len(p) - 1
    NEG = -np.inf
    G = np.full((max_budget + 1, max_depth + 1, B + 1), NEG)
    split = np.zeros((max_budget + 1, max_depth + 1, B + 1), dtype=np.int64)     # y* (nodes kept by the first b-1 kids)
    G[1, 1:, 0] = 1.0
    F = np.full((max_budget + 1, max_depth + 1), NEG)
    F[1, 1:] = 1.0
    for l in range(2, max_depth + 1):
        for m in range(2, max_budget + 1):
            G[m, l, 1] = 1.0 + p[1] * F[m - 1, l - 1]
            split[m, l, 1] = 1
            for b in range(2, B + 1):
                ys = np.arange(1, m)
                cand = G[ys, l, b - 1] + p[b] * F[m - ys, l - 1]
                j = int(np.argmax(cand))                                         # first maximum, like the reference's `>`
                G[m, l, b] = cand[j]
                split[m, l, b] = ys[j]
            F[m, l] = G[m, l].max()
    return F, (G, split)


def expand_tree(back, m, l):
    """Branch lists (level by level, parents in creation order) of the optimal (m, l) tree."""
    G, split = back

    def kids_of(mm, ll, b):
        out = []
        while b >= 1:                                                            # peel the last child off
            y = int(split[mm, ll, b]) if b > 1 else 1
            sub_m = mm - y
            sub_b = int(np.argmax(G[sub_m, ll - 1]))
            out.append((sub_m, ll - 1, sub_b))
            mm, b = y, b - 1
        return out[::-1]

    level = [(m, l, int(np.argmax(G[m, l])))]
    branches = []
    while level:
        branches.append([s[2] for s in level])
        nxt = []
        for (mm, ll, b) in level:
            nxt.extend(kids_of(mm, ll, b))
        level = nxt
    return branches


def choose_shape(F, valid_budgets, draft_time, target_times):
    """The reference's selection rule (tree_search.py:57-70): minimise (depth * draft_time + target_time) / E[accepted]."""
    best = (np.inf, None)
    for b, tt in zip(valid_budgets, target_times):
        for d in range(F.shape[1]):
            if F[b, d] <= 0 or not np.isfinite(F[b, d]):
                continue
            x = (d * draft_time + tt) / F[b, d]
            if x < best[0]:
                best = (x, (b, d))
    return best[1]


def build_grow_map(size=512, max_depth=16, acceptance=None):
    F, back = search_tree(acceptance or DEFAULT_ACCEPTANCE, size, max_depth)
    return grow_map_from_branches(expand_tree(back, size, max_depth))


def load_grow_map(spec, cache_dir=None):
    """``spec``: a path to a grow map saved with torch.save (the reference's tree/512.pt format), a JSON file
    holding ``{"branches": [...]}``, or an int / digit string = build (and cache) a tree of that many nodes."""
    if isinstance(spec, str) and os.path.exists(spec):
        if spec.endswith(".json"):
            with open(spec) as f:
                return grow_map_from_branches(json.load(f)["branches"])
        return torch.load(spec)
    size = int(spec)
    cache_dir = cache_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tree")
    path = os.path.join(cache_dir, f"{size}.json")
    if os.path.exists(path):
        with open(path) as f:
            return grow_map_from_branches(json.load(f)["branches"])
    gm = build_grow_map(size=size, max_depth=min(16, size))     # 16 levels like the reference's tree/512.pt
    try:                                   # every torchrun rank may get here at once: write aside, publish atomically
        os.makedirs(cache_dir, exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump({"branches": gm["branches"], "size": gm["size"]}, f)
        os.replace(tmp, path)
    except OSError:
        pass
    return gm
