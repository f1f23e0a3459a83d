"""Makes the reference's own entry points (``main.py``, ``train_model.py``,
``evaluate.py``, ``models/ReaRev/rearev.py``) use the MI355X modules without editing
a single reference file.

Two ways:

1. ``install()`` BEFORE the reference imports its model code: registers this package's
   modules under the names the reference imports
   (``modules.kg_reasoning.reasongnn``, ``modules.kg_reasoning.base_gnn``,
   ``modules.layer_init``, ``modules.query_update``), so ``from modules.kg_reasoning.reasongnn import
   ReasonGNNLayer`` (rearev.py:8) resolves to the HIP-backed class.
2. ``swap(model)`` AFTER construction: replaces ``model.reasoning`` / ``model.type_layer`` /
   ``model.reform{j}`` of an existing ReaRev instance (or the ``NSMLayer`` / ``NSMLayer_back`` layers and the
   TypeLayer of an NSM instance), carrying the parameters over.
"""
from __future__ import annotations

import importlib
import sys

_TARGETS = {
    "modules.kg_reasoning.reasongnn": "gnnrag_amd.modules.kg_reasoning.reasongnn",
    "modules.kg_reasoning.base_gnn": "gnnrag_amd.modules.kg_reasoning.base_gnn",
    "modules.kg_reasoning.nsm_gnn": "gnnrag_amd.modules.kg_reasoning.nsm_gnn",
    "modules.layer_init": "gnnrag_amd.modules.layer_init",
    "modules.query_update": "gnnrag_amd.modules.query_update",
}


def install() -> None:
    already = [n for n in _TARGETS if n in sys.modules and
               not getattr(sys.modules[n], "__name__", "").startswith("gnnrag_amd")]
    if already:
        raise RuntimeError("install() must run before the reference imports %s "
                           "(use swap(model) on an existing model instead)" % already)
    for ref_name, our_name in _TARGETS.items():
        sys.modules[ref_name] = importlib.import_module(our_name)


def host_cpu_budget(cgroup_root: str = "/sys/fs/cgroup") -> int:
    """CPU cores this process may actually use: ``os.cpu_count()`` capped by the cgroup's CFS quota (``cpu.max`` of
    cgroup v2, ``cpu.cfs_quota_us / cpu.cfs_period_us`` of v1).  A container on a 256-thread host is typically given a
    fraction of it, and ``os.cpu_count()`` (hence torch's / OpenMP's default thread count) does not know."""
    import os
    n = os.cpu_count() or 1
    try:
        with open(cgroup_root + "/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open(cgroup_root + "/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open(cgroup_root + "/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def host_thread_limit(budget: int | None = None, environ=None) -> int:
    """Threads one GPU-driving process may use: ``min(8, budget // (2 * local ranks))``, at least 1, and never more than an
    ``OMP_NUM_THREADS`` the launcher already set (``torch.distributed.run`` sets 1 when it starts more than one rank).
    ``local ranks`` = ``LOCAL_WORLD_SIZE`` (else ``WORLD_SIZE``) of the environment: the ranks of one node share ONE CPU
    quota, so eight ranks with eight spinning workers each would bring the throttling back on exactly the multi-GPU path."""
    import os
    env = os.environ if environ is None else environ
    if budget is None:
        budget = host_cpu_budget()

    def _int(name):
        v = str(env.get(name, "")).strip()
        return int(v) if v.isdigit() and int(v) > 0 else None

    local = _int("LOCAL_WORLD_SIZE") or _int("WORLD_SIZE") or 1
    n = max(1, min(8, int(budget) // (2 * local)))
    omp = _int("OMP_NUM_THREADS")
    return min(n, omp) if omp else n


def limit_host_threads(n: int | None = None) -> int:
    """Intra-op CPU threads of the process that DRIVES the GPU: ``host_thread_limit()`` unless given -
    ``min(8, host_cpu_budget() // (2 * local ranks))``, capped by an explicit ``OMP_NUM_THREADS``.
    The GPU path leaves the host a few hundred small launches and a handful of [B, N] conversions per forward - nothing to
    parallelise - but torch starts one OpenMP worker per visible hardware thread, and the workers spin after every
    parallel region.  On the MI355X box (256 hardware threads visible, a 16-core CFS quota) that exhausts the quota: the
    kernel parks EVERY thread of the process for the rest of the 100 ms period, the launching thread included.  Measured
    through the unmodified main.py (profiles/r05i_forward_host_time.txt): a 64-question forward 12-22 ms -> 4.8 ms of
    host time, one 68-78 ms pause every few batches -> none, ``nr_throttled`` 17 -> 0.  Returns the thread count set."""
    import torch
    if n is None:
        n = host_thread_limit()
    torch.set_num_threads(int(n))
    return int(n)


def freeze_loaded_data() -> int:
    """Call once after the reference has loaded its data and built its model (and before the evaluation / training
    loop): ``gc.collect(); gc.freeze()``.  The reference's loaders (dataset_load.py:24-160) keep every question of every
    split as Python lists / dicts / numpy rows - millions of container objects - and CPython's cyclic collector walks
    all of them at every full collection (1.2 million objects for the 680 staged questions; WebQSP has 4 700).  Frozen
    objects are never walked again; nothing is leaked that the run would have freed.  On the staged data the collector
    turned out NOT to be what paused the forward (two collections of 0.04 ms inside 8 forwards - the pauses were CPU-quota
    throttling, see limit_host_threads); the freeze is kept as the cheap insurance it is for the full-size datasets.
    Returns the number of frozen objects."""
    import gc
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()


def uninstall() -> None:
    for ref_name in _TARGETS:
        m = sys.modules.get(ref_name)
        if m is not None and m.__name__.startswith("gnnrag_amd"):
            del sys.modules[ref_name]


def cache_rel_features(model):
    """``ReaRev.get_rel_feature`` (rearev.py:91-111) re-encodes the relation vocabulary on EVERY forward: a linear
    over the relation embeddings, or - with ``--relation_word_emb True`` - ``question_emb`` + ``AttnEncoder`` over the
    LM states of all relation texts ``[R1, W, dim]`` (18.8 GFLOP per batch for WebQSP's 6105 relations, as much as the
    GNN itself).  Neither depends on the batch.  In evaluation (no grad, eval mode) the result is kept and reused
    until a parameter it depends on changes (checked by storage and version of every model parameter); training calls
    always recompute.  The reference file is untouched: the bound method is wrapped on the instance."""
    import torch
    inner = model.get_rel_feature
    if getattr(inner, "_gnnrag_cached", False):
        return model
    state = {"key": None, "value": None}

    def get_rel_feature():
        if torch.is_grad_enabled() or model.training:
            return inner()
        key = tuple((p.data_ptr(), p._version) for p in model.parameters())
        if state["key"] != key:
            state["value"], state["key"] = inner(), key
        return state["value"]

    get_rel_feature._gnnrag_cached = True
    model.get_rel_feature = get_rel_feature
    return model


def _swap_nsm(model, args: dict):
    """``models/NSM/nsm.py``: ``reasoning`` / ``reasoning2`` (NSMLayer) and ``reasoning_back`` (NSMLayer_back)."""
    from .modules.kg_reasoning import nsm_gnn
    num_relation = getattr(model, "num_relation", None)
    for name in ("reasoning", "reasoning2", "reasoning_back"):
        old = getattr(model, name, None)
        if old is None:
            continue
        cls = nsm_gnn.NSMLayer_back if type(old).__name__ == "NSMLayer_back" else nsm_gnn.NSMLayer
        # old.num_relation is overwritten by init_reason (nsm_gnn.py:44); the constructor value is kept by BaseModel
        new = cls(args, old.num_entity, num_relation if num_relation is not None else old.num_relation, old.entity_dim)
        new.load_state_dict(old.state_dict(), strict=True)
        new.to(next(old.parameters()).device)
        new.train(old.training)
        setattr(model, name, new)


def swap(model, args: dict):
    """Replaces the reasoning layer(s) (and TypeLayer, if present) of a constructed ReaRev or NSM model."""
    from .modules.kg_reasoning.reasongnn import ReasonGNNLayer
    from .modules.layer_init import TypeLayer
    old = model.reasoning
    if type(old).__name__ in ("NSMLayer", "NSMLayer_back"):
        _swap_nsm(model, args)
        if getattr(model, "type_layer", None) is not None:
            tl_old = model.type_layer
            tl = TypeLayer(tl_old.in_features, tl_old.out_features, tl_old.linear_drop, tl_old.device, tl_old.norm_rel)
            tl.load_state_dict(tl_old.state_dict(), strict=True)
            tl.to(next(tl_old.parameters()).device)
            model.type_layer = tl
        return model
    # old.num_relation is overwritten by init_reason (reasongnn.py:55); the constructor value,
    # which sizes pos_emb, is kept by BaseModel (base_model.py:21)
    num_relation = getattr(model, "num_relation", old.num_relation)
    new = ReasonGNNLayer(args, old.num_entity, num_relation, old.entity_dim, old.alg)
    new.load_state_dict(old.state_dict(), strict=True)
    new.to(next(old.parameters()).device)
    new.train(old.training)
    model.reasoning = new
    if getattr(model, "type_layer", None) is not None:
        tl_old = model.type_layer
        tl = TypeLayer(tl_old.in_features, tl_old.out_features, tl_old.linear_drop, tl_old.device,
                       tl_old.norm_rel)
        tl.load_state_dict(tl_old.state_dict(), strict=True)
        tl.to(next(tl_old.parameters()).device)
        model.type_layer = tl
    from .modules.query_update import QueryReform
    j = 0
    while getattr(model, "reform" + str(j), None) is not None:       # rearev.py:46-47
        old_r = getattr(model, "reform" + str(j))
        new_r = QueryReform(old_r.q_ent_attn.in_features)
        new_r.load_state_dict(old_r.state_dict(), strict=True)
        new_r.to(next(old_r.parameters()).device)
        new_r.train(old_r.training)
        setattr(model, "reform" + str(j), new_r)
        j += 1
    if hasattr(model, "get_rel_feature"):
        cache_rel_features(model)
    swap_lstm(model)
    return model


def swap_lstm(model) -> int:
    """The question encoder's ``nn.LSTM`` (lstm_encoder.py:27-30) -> ``HipLSTM`` (same parameters, shared); works on a model
    built either way (install() or the reference's own modules).  Returns the number of LSTMs replaced."""
    from .modules.question_encoding.lstm import swap_lstm as _swap
    return _swap(model)
