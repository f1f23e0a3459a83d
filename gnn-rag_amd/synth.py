"""Synthetic Freebase-shaped question subgraphs in the reference's batch format.

The tuple produced by :func:`make_edge_tuple` has exactly the layout that
``BasicDataLoader._build_fact_mat`` hands to the model (reference
``gnn/dataset_load.py:473-527``): heads/tails already offset by ``i * N`` per
question, facts of one question contiguous, optional self-loop facts with
relation id ``num_kb_relation - 1`` appended per question, ``weight_list`` =
1/outdeg(head) and ``weight_rel_list`` = 1/count(head, rel).

Shapes follow SURVEY.md section 8(d) ("Synthetic inputs (C2, pinned)").
numpy only - this module is used by tests, bench.py and the oracle alike.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

SEED_DEFAULT = 19960626  # reference default seed, gnn/parsing.py:48


@dataclass
class GraphConfig:
    """One synthetic workload (names follow BASELINE.json configs)."""
    name: str = "C2"
    B: int = 64            # questions per batch
    N: int = 2000          # max_local_entity (padded nodes per question)
    E: int = 10000         # typed edges per question
    R: int = 600           # distinct KB relation types (self-loop id = R)
    D: int = 200           # entity_dim
    I: int = 2             # num_ins
    L: int = 3             # num_gnn
    T: int = 1             # num_iter (outer iterations; dist reset each)
    zipf_heads: bool = True
    self_loop: bool = True
    normalized_gnn: bool = False
    pos_emb: bool = False
    n_real_min: Optional[int] = None   # ragged: real nodes per question in [n_real_min, N]
    rel_per_question: Optional[int] = None   # each question draws its relations from its own subset of
                                             # this many ids out of R (a Freebase subgraph touches a few
                                             # hundred of the ~6k relations); None = uniform over R
    seed: int = SEED_DEFAULT

    @property
    def num_kb_relation(self) -> int:      # incl. the self-loop relation
        return self.R + 1

    @property
    def R1(self) -> int:                   # rows of the relation feature tables
        return self.R + 2                  # dataset_load.py:413 (num_kb_relation + 1)


CONFIGS = {
    # BASELINE.json configs[1]: the config the metric is quoted on
    "C2": GraphConfig(name="C2", B=64, N=2000, E=10000, R=600, D=200, I=2, L=3),
    # configs[3]: CWQ-like, 4 layers, num_ins 3 (scripts/rearev_cwq.sh:14)
    "C4": GraphConfig(name="C4", B=32, N=5000, E=30000, R=600, D=200, I=3, L=4),
    # configs[4] per-GPU share (32 questions per GPU)
    "C5": GraphConfig(name="C5", B=32, N=20000, E=200000, R=6000, D=200, I=2, L=3),
    # configs[0]/[2]-shaped: WebQSP-like ragged questions, released-ckpt dims (D=50)
    "C1": GraphConfig(name="C1", B=1, N=2000, E=6000, R=600, D=50, I=2, L=3, T=3,
                      n_real_min=50),
    "C3": GraphConfig(name="C3", B=32, N=2000, E=6000, R=600, D=50, I=2, L=3, T=3,
                      n_real_min=50),
    # C2 with the full Freebase relation vocabulary of WebQSP (6105 relations), a few hundred per question
    "C2fb": GraphConfig(name="C2fb", B=64, N=2000, E=10000, R=6105, D=200, I=2, L=3, rel_per_question=300),
    # C2 with uniformly drawn heads (no hubs): what the degree skew costs the walks (not a BASELINE config)
    "C2u": GraphConfig(name="C2u", B=64, N=2000, E=10000, R=600, D=200, I=2, L=3, zipf_heads=False),
    # small cases for parity tests
    "tiny": GraphConfig(name="tiny", B=3, N=48, E=150, R=11, D=200, I=2, L=3),
    "tinyfb": GraphConfig(name="tinyfb", B=5, N=64, E=300, R=1500, D=200, I=2, L=3, rel_per_question=40,
                          n_real_min=0),
    "tiny50": GraphConfig(name="tiny50", B=4, N=40, E=90, R=7, D=50, I=3, L=2,
                          normalized_gnn=True, pos_emb=True, n_real_min=5),
}


@dataclass
class Batch:
    """Everything one hot-path invocation needs (host side, numpy)."""
    cfg: GraphConfig
    local_entity: np.ndarray      # int64 [B,N] global ids, pad = num_entity
    query_entities: np.ndarray    # float64 [B,N] 1.0 at seeds
    seed_dist: np.ndarray         # float64 [B,N]
    edge_tuple: tuple             # (heads, rels, tails, batch_ids, fact_ids, weight_list, weight_rel_list)
    num_entity: int
    n_real: np.ndarray            # int64 [B] real nodes per question
    extras: dict = field(default_factory=dict)

    @property
    def F(self) -> int:
        return int(len(self.edge_tuple[0]))


def make_edge_tuple(cfg: GraphConfig, rng: np.random.Generator, n_real: np.ndarray):
    """Restates the semantics of ``_build_fact_mat`` (dataset_load.py:473-527)
    for synthetic questions: per question typed edges (permuted), then
    self-loops over the real entities; then the two weight lists."""
    heads, rels, tails, bids = [], [], [], []
    for i in range(cfg.B):
        n = int(n_real[i])
        e = cfg.E if n == cfg.N else max(1, int(cfg.E * n / cfg.N))
        if n == 0:
            e = 0
        if cfg.zipf_heads and n > 0:
            h = (rng.zipf(1.6, size=e) % n).astype(np.int64)
        else:
            h = rng.integers(0, max(n, 1), size=e, dtype=np.int64)
        t = rng.integers(0, max(n, 1), size=e, dtype=np.int64)
        if cfg.rel_per_question:
            own = rng.choice(cfg.R, size=min(cfg.rel_per_question, cfg.R), replace=False)
            r = own[(rng.zipf(1.3, size=e) - 1) % len(own)].astype(np.int64)
        else:
            r = rng.integers(0, cfg.R, size=e, dtype=np.int64)
        perm = rng.permutation(e)                      # dataset_load.py:489-490
        off = i * cfg.N                                # dataset_load.py:483
        heads.append(h[perm] + off)
        tails.append(t[perm] + off)
        rels.append(r[perm])
        bids.append(np.full(e, i, dtype=np.int64))
        if cfg.self_loop:                              # dataset_load.py:499-506
            ent = np.arange(n, dtype=np.int64) + off
            heads.append(ent)
            tails.append(ent)
            rels.append(np.full(n, cfg.num_kb_relation - 1, dtype=np.int64))
            bids.append(np.full(n, i, dtype=np.int64))
    heads = np.concatenate(heads) if heads else np.zeros(0, np.int64)
    rels = np.concatenate(rels) if rels else np.zeros(0, np.int64)
    tails = np.concatenate(tails) if tails else np.zeros(0, np.int64)
    bids = np.concatenate(bids) if bids else np.zeros(0, np.int64)
    fact_ids = np.arange(len(heads), dtype=np.int64)
    weight_list, weight_rel_list = edge_weights(heads, rels, cfg.R1)
    return (heads, rels, tails, bids, fact_ids, weight_list, weight_rel_list)


def edge_weights(heads: np.ndarray, rels: np.ndarray, R1: int):
    """1/outdeg(head) and 1/count(head, rel) (dataset_load.py:509-517), vectorised.
    Returned as Python lists of float, the reference's own container type."""
    if len(heads) == 0:
        return [], []
    hc = np.bincount(heads)
    wl = (1.0 / hc[heads]).tolist()
    key = heads * np.int64(R1) + rels
    _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    wrl = (1.0 / cnt[inv]).tolist()
    return wl, wrl


def make_batch(cfg: GraphConfig, seed: Optional[int] = None) -> Batch:
    rng = np.random.default_rng(cfg.seed if seed is None else seed)
    num_entity = max(10 * cfg.N, 1000)
    if cfg.n_real_min is None:
        n_real = np.full(cfg.B, cfg.N, dtype=np.int64)
    else:
        n_real = rng.integers(cfg.n_real_min, cfg.N + 1, size=cfg.B, dtype=np.int64)
    local_entity = np.full((cfg.B, cfg.N), num_entity, dtype=np.int64)
    for i in range(cfg.B):
        local_entity[i, : n_real[i]] = rng.integers(0, num_entity, size=n_real[i])
    query_entities = np.zeros((cfg.B, cfg.N), dtype=np.float64)
    seed_dist = np.zeros((cfg.B, cfg.N), dtype=np.float64)
    # node 0 of each question is its seed; on WebQSP the seed slot keeps the pad id,
    # so seeds are masked out of the answer softmax (dataset_load.py:249-257)
    has = n_real > 0
    query_entities[has, 0] = 1.0
    seed_dist[has, 0] = 1.0
    local_entity[has, 0] = num_entity
    edge_tuple = make_edge_tuple(cfg, rng, n_real)
    return Batch(cfg=cfg, local_entity=local_entity, query_entities=query_entities,
                 seed_dist=seed_dist, edge_tuple=edge_tuple, num_entity=num_entity,
                 n_real=n_real)


def make_features(cfg: GraphConfig, seed: Optional[int] = None) -> dict:
    """Dense inputs of the layer stack: h0, relation tables, instructions (fp32)."""
    rng = np.random.default_rng((cfg.seed if seed is None else seed) + 1)
    f32 = np.float32
    return {
        "h0": (0.1 * rng.standard_normal((cfg.B, cfg.N, cfg.D))).astype(f32),
        "rel_features": (0.3 * rng.standard_normal((cfg.R1, cfg.D))).astype(f32),
        "rel_features_inv": (0.3 * rng.standard_normal((cfg.R1, cfg.D))).astype(f32),
        # one instruction tensor per outer iteration (QueryReform output stand-in)
        "ins": (0.3 * rng.standard_normal((cfg.T, cfg.B, cfg.I, cfg.D))).astype(f32),
    }


def make_layer_params(cfg: GraphConfig, seed: Optional[int] = None) -> dict:
    """Parameters named as in the reference ``state_dict`` (reasongnn.py:26-44),
    drawn with nn.Linear's default init bounds (uniform +-1/sqrt(fan_in))."""
    rng = np.random.default_rng((cfg.seed if seed is None else seed) + 2)
    f32 = np.float32
    D, I = cfg.D, cfg.I

    def lin(out_f, in_f):
        k = 1.0 / np.sqrt(in_f)
        return (rng.uniform(-k, k, size=(out_f, in_f)).astype(f32),
                rng.uniform(-k, k, size=(out_f,)).astype(f32))

    p = {}
    p["score_func.weight"], p["score_func.bias"] = lin(1, D)
    p["glob_lin.weight"], p["glob_lin.bias"] = lin(D, D)
    p["lin.weight"], p["lin.bias"] = lin(D, 2 * D)
    for s in range(cfg.L):
        p[f"rel_linear{s}.weight"], p[f"rel_linear{s}.bias"] = lin(D, D)
        p[f"e2e_linear{s}.weight"], p[f"e2e_linear{s}.bias"] = lin(D, (2 * I + 1) * D)
        if cfg.pos_emb:
            p[f"pos_emb{s}.weight"] = rng.standard_normal((cfg.num_kb_relation, D)).astype(f32)
            p[f"pos_emb_inv{s}.weight"] = rng.standard_normal((cfg.num_kb_relation, D)).astype(f32)
    p["lin_m.weight"], p["lin_m.bias"] = lin(D, I * D)
    # TypeLayer (layer_init.py:19)
    p["type_layer.kb_self_linear.weight"], p["type_layer.kb_self_linear.bias"] = lin(D, D)
    return p
