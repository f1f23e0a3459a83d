"""Vectorised replacement for ``BasicDataLoader._build_fact_mat``.

The reference (``gnn/dataset_load.py:473-527``) grows four arrays with ``np.append`` inside the
per-question loop (every append copies everything gathered so far: quadratic in the batch) and
computes the two weight lists with Python ``Counter`` objects and list comprehensions over all
facts - 1.2 s per 64-question batch at C2 size, three orders of magnitude more than the GPU
forward.  Here the per-question pieces are collected in lists and concatenated once, and the
weights come from ``np.unique`` / ``np.bincount``.

Drop-in: same arguments, same 7-tuple (same dtypes and container types: int arrays + two Python
lists of float), and - because the per-question ``np.random.permutation`` calls are made in the
same order with the same sizes - the *same* tuple as the reference for the same numpy RNG state
(``tests/test_fact_mat.py`` checks that against the live reference).

    from gnnrag_amd.data.fact_mat import patch_loader
    patch_loader(dataset["test"])          # loader._build_fact_mat is now the fast one
"""
from __future__ import annotations

import types

import os

import numpy as np


def build_fact_mat(loader, sample_ids, fact_dropout):
    """``loader`` is a reference ``BasicDataLoader`` (or anything with its attributes
    ``max_local_entity``, ``data_eff``, ``kb_adj_mats`` / ``create_kb_adj_mats``,
    ``use_self_loop``, ``global2local_entity_maps``, ``num_kb_relation``)."""
    N = loader.max_local_entity
    heads, rels, tails, bids = [], [], [], []
    for i, sample_id in enumerate(sample_ids):
        index_bias = i * N                                                    # dataset_load.py:483
        if loader.data_eff:
            head_list, rel_list, tail_list = loader.create_kb_adj_mats(sample_id)
        else:
            head_list, rel_list, tail_list = loader.kb_adj_mats[sample_id]
        num_fact = len(head_list)
        num_keep_fact = int(np.floor(num_fact * (1 - fact_dropout)))          # :488
        mask_index = np.random.permutation(num_fact)[:num_keep_fact]          # :489-490 (same RNG draws)
        heads.append(head_list[mask_index] + index_bias)
        tails.append(tail_list[mask_index] + index_bias)
        rels.append(rel_list[mask_index])
        bids.append(np.full(len(mask_index), i, dtype=int))
        if loader.use_self_loop:                                              # :499-506
            num_ent_now = len(loader.global2local_entity_maps[sample_id])
            ent = np.arange(num_ent_now, dtype=int) + index_bias
            heads.append(ent)
            tails.append(ent)
            rels.append(np.full(num_ent_now, loader.num_kb_relation - 1, dtype=int))
            bids.append(np.full(num_ent_now, i, dtype=int))

    def cat(parts):
        return np.concatenate(parts) if parts else np.array([], dtype=int)

    batch_heads, batch_rels, batch_tails, batch_ids = cat(heads), cat(rels), cat(tails), cat(bids)
    fact_ids = np.arange(len(batch_heads), dtype=int)                         # :507
    if len(batch_heads):
        head_count = np.bincount(batch_heads)                                 # :509-511 (Counter over heads)
        weight_list = (1.0 / head_count[batch_heads]).tolist()
        key = batch_heads.astype(np.int64) * (int(batch_rels.max()) + 1) + batch_rels
        _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)  # :513-517 (Counter over pairs)
        weight_rel_list = (1.0 / cnt[inv]).tolist()
    else:
        weight_list, weight_rel_list = [], []
    return batch_heads, batch_rels, batch_tails, batch_ids, fact_ids, weight_list, weight_rel_list


class _QuestionFacts:
    """One question's facts: ``[0]`` the [3, F] int32 id block, ``[1]`` / ``[2]`` the two per-fact weights
    (``dataset_load.py:509-517``) - computed when first asked for: only ``normalized_gnn`` / ``norm_rel`` models read them,
    and the pair count behind the second is a sort over the question's facts."""

    __slots__ = ("blk", "_w", "_wr", "nrel")

    def __init__(self, blk):
        self.blk, self._w, self._wr = blk, None, None
        # distinct relation ids among the question's facts (self loops included): what the device structure build counts
        # per question - known here, so a batch's build can be told the counts instead of waiting for them
        self.nrel = int(len(np.unique(blk[1]))) if blk.shape[1] else 0

    def __getitem__(self, k):
        if k == 0:
            return self.blk
        h, r = self.blk[0], self.blk[1]
        if k == 1:
            if self._w is None:
                self._w = 1.0 / np.bincount(h)[h] if len(h) else np.zeros(0)            # :509-511
            return self._w
        if k == 2:
            if self._wr is None:
                if len(h):
                    key = h.astype(np.int64) * (int(r.max()) + 1) + r
                    _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
                    self._wr = 1.0 / cnt[inv]                                              # :513-517
                else:
                    self._wr = np.zeros(0)
            return self._wr
        raise IndexError(k)


class FactCache:
    """Per-question fact arrays built ONCE: typed edges followed by the self
    loops, local node ids, with both per-fact weights - everything ``_build_fact_mat`` recomputes for
    every batch.  A batch is then a concatenation with node offsets.

    Used only when ``fact_dropout == 0`` (evaluation; ``evaluate.py:158`` always passes 0.0): the
    reference still draws a random permutation of each question's facts there
    (``dataset_load.py:489-490``), which only reorders the facts inside the tuple - every consumer is a
    sum over facts, so the results agree to fp32 summation order.  The cached path keeps the stored
    order (= the reference's for the identity permutation) and does not touch numpy's RNG.

    Containers: the three id arrays are rows of ONE int32 block (uploaded by ``ops.CsrPlan`` without a
    copy); the two weight "lists" are float64 arrays - every consumer in the reference
    (``torch.FloatTensor(weight_list)``, ``base_gnn.py:38-40``, ``layer_init.py:39-40``) takes either."""

    def __init__(self, loader, max_questions: int = 200000):
        self.loader = loader
        self._q = {}
        self.max_questions = max_questions      # questions kept; beyond that they are rebuilt per batch

    def _question(self, sample_id):
        q = self._q.get(sample_id)
        if q is None:
            ld = self.loader
            if ld.data_eff:
                h, r, t = ld.create_kb_adj_mats(sample_id)
            else:
                h, r, t = ld.kb_adj_mats[sample_id]
            h, r, t = (np.asarray(x, dtype=np.int32) for x in (h, r, t))
            if ld.use_self_loop:                                              # dataset_load.py:499-506
                ent = np.arange(len(ld.global2local_entity_maps[sample_id]), dtype=np.int32)
                h = np.concatenate([h, ent])
                t = np.concatenate([t, ent])
                r = np.concatenate([r, np.full(len(ent), ld.num_kb_relation - 1, dtype=np.int32)])
            q = _QuestionFacts(np.stack([h, r, t]))
            if len(self._q) < self.max_questions:
                self._q[sample_id] = q
        return q

    def batch(self, sample_ids):
        N = self.loader.max_local_entity
        parts = [self._question(int(s)) for s in sample_ids]
        sizes = np.array([p[0].shape[1] for p in parts], dtype=np.int64)
        F = int(sizes.sum())
        hrt = np.concatenate([p[0] for p in parts], axis=1) if parts else np.zeros((3, 0), np.int32)
        batch_ids = np.repeat(np.arange(len(parts), dtype=np.int64), sizes)
        off = (batch_ids * N).astype(np.int32)                                # dataset_load.py:483
        hrt[0] += off
        hrt[2] += off
        cat = lambda k: np.concatenate([p[k] for p in parts]) if parts else np.zeros(0)
        return hrt[0], hrt[1], hrt[2], batch_ids, np.arange(F, dtype=np.int64), cat(1), cat(2)


class BatchFacts:
    """``kb_adj_mat`` of a batch served from a :class:`DeviceFactCache`: behaves like the reference's 7-tuple
    (``dataset_load.py:527``) for everything the MI355X modules read, but the three id arrays are rows of ONE int32
    block that already lives on the GPU (``hrt_device`` [3, F]; items 0..2 are views of it) and the host-side members
    (batch ids, fact ids, the two weight lists) are built only if somebody asks for them (``normalized_gnn`` /
    ``norm_rel``).  The reference's own modules cannot consume it (they build ``torch.LongTensor`` from numpy arrays):
    it is handed out only by ``patch_loader(..., device=...)``, i.e. next to ``install.swap``-ed modules."""

    def __init__(self, hrt_device, sizes, parts, N, plans=None, make_hrt=None, checked=False, checked_r1=None, owner=None):
        self._hrt, self._make_hrt = hrt_device, make_hrt
        self._sizes, self._parts, self._N = sizes, parts, N
        # what the host range check of the cache covered: node ids < N, relation ids < checked_r1.  plan_for passes the
        # relation counts on (no-wait build) only for a structure of exactly this N and at least this many table rows;
        # `owner` (the cache) gets ONE deferred device-side validation (CsrPlan.status) on its first no-wait build
        self.checked_n, self.checked_r1, self.owner = (N if checked else None), checked_r1, owner
        # (rel_total, rel_max) of the batch for ops.CsrPlan(rel_counts=): only when every question's ids were range-checked
        # on the host as it was cached (the build then skips its wait AND its read-back of the device-side validation)
        self.rel_counts = ((sum(p.nrel for p in parts), max([p.nrel for p in parts] or [0])) if checked and parts else None)
        self._lazy = {}
        # DeviceStructureCache: the questions' cached single-question structures (ops.CsrPlan, B = 1), in batch order -
        # the modules then assemble the batch structure by concatenation (ops.CsrPlan.concat) instead of sorting
        self.plans = plans

    @property
    def hrt_device(self):
        """[3, F] int32 id block on the GPU; built on first access when the batch came from cached structures."""
        if self._hrt is None:
            self._hrt = self._make_hrt()
        return self._hrt

    def __len__(self):
        return 7

    def _host(self, k):
        if k not in self._lazy:
            F = int(self._sizes.sum())
            if k == 3:
                v = np.repeat(np.arange(len(self._parts), dtype=np.int64), self._sizes)
            elif k == 4:
                v = np.arange(F, dtype=np.int64)
            else:
                v = np.concatenate([p[k - 4] for p in self._parts]) if self._parts else np.zeros(0)
            self._lazy[k] = v
        return self._lazy[k]

    def __getitem__(self, k):
        if isinstance(k, slice):
            return tuple(self[i] for i in range(7)[k])
        if k < 0:
            k += 7
        if k in (0, 1, 2):
            return self.hrt_device[k]
        if 3 <= k < 7:
            return self._host(k)
        raise IndexError(k)

    def __iter__(self):
        return (self[i] for i in range(7))

    def shard(self, lo: int, hi: int) -> "BatchFacts":
        """Facts of questions [lo, hi) as a self-contained batch (node ids re-based by ``lo * N``), sliced ON THE
        DEVICE - what ``shard.shard_edge_tuple`` hands a rank when the loader serves device-resident tuples."""
        a, b = int(self._sizes[:lo].sum()), int(self._sizes[:hi].sum())
        if self.plans is not None:               # cached structures shard by slicing the list
            parts, sizes, N, plans = self._parts[lo:hi], self._sizes[lo:hi], self._N, self.plans[lo:hi]
            return BatchFacts(None, sizes, parts, N, plans=plans, make_hrt=lambda: _cat_blocks([p._hrt[:, : p.F] for p in plans], sizes, N))
        hrt = self.hrt_device[:, a:b].clone()
        if lo:
            hrt[0] -= lo * self._N
            hrt[2] -= lo * self._N
        return BatchFacts(hrt, self._sizes[lo:hi], self._parts[lo:hi], self._N, checked=self.rel_counts is not None,
                          checked_r1=self.checked_r1, owner=self.owner)


class ShardedFacts:
    """``kb_adj_mat`` of a batch of which THIS RANK built only its own questions (``patch_loader(..., shard=(rank,
    world))``): under question sharding (``shard.shard_model``) a rank's forward reads the facts of its contiguous
    question range only, so building the global batch's tuple on every rank and dropping (world - 1) / world of it is
    wasted host time.  ``facts_per_question`` (all B questions: known from the loader without building
    anything) and ``ranges`` (the fact-balanced split every rank derives identically) are what ``shard.shard_ranges``
    reads; ``shard(lo, hi)`` hands out the local tuple for this rank's own range and refuses any other.  Anything that
    indexes it like the reference's 7-tuple is told what it is holding."""

    def __init__(self, local, facts_per_question, ranges, rank):
        self.local, self.facts_per_question, self.ranges, self.rank = local, facts_per_question, ranges, rank

    def __len__(self):
        return 7

    def shard(self, lo: int, hi: int):
        if (lo, hi) != tuple(self.ranges[self.rank]):
            raise ValueError("rank-local batch: rank %d built questions [%d, %d) only, [%d, %d) was asked for"
                             % ((self.rank,) + tuple(self.ranges[self.rank]) + (lo, hi)))
        return self.local

    def __getitem__(self, k):
        raise TypeError("rank-local batch (fact_mat.ShardedFacts): only this rank's questions were built - run the model "
                        "through shard.shard_model, or patch the loader without shard=")

    __iter__ = None


def _cat_blocks(blocks, sizes, N):
    """Per-question [3, F_g] int32 blocks (question-local node ids) -> the batch's [3, F] block with node offsets."""
    import torch
    if not blocks:
        return torch.zeros((3, 0), dtype=torch.int32)
    dev = blocks[0].device
    hrt = torch.cat(blocks, dim=1)
    off = torch.repeat_interleave(torch.arange(len(blocks), device=dev, dtype=torch.int32) * N,
                                  torch.from_numpy(np.asarray(sizes, dtype=np.int64)).to(dev, non_blocking=True),
                                  output_size=int(np.sum(sizes)))        # size given: no device->host sync
    hrt[0] += off
    hrt[2] += off
    return hrt


class DeviceFactCache(FactCache):
    """:class:`FactCache` whose per-question id blocks live ON THE GPU ("cached per-question
    int32 [structure] built once at load time, batch = concatenation with offsets"): a question's [3, F_g] int32 block
    is uploaded the first time it is used; a batch is one device-side concatenation plus the node offsets - no host
    concatenation of the facts and no per-batch PCIe transfer of them (4.7 ms + 9 MB per C2 batch before)."""

    def __init__(self, loader, device, max_questions: int = 200000):
        super().__init__(loader, max_questions)
        import torch
        self.device = torch.device(device)
        self._dev = {}

    def batch(self, sample_ids):
        import torch
        N = self.loader.max_local_entity
        parts, blocks = [], []
        for s_ in sample_ids:
            s_ = int(s_)
            q = self._question(s_)
            d = self._dev.get(s_)
            if d is None:
                blk = q[0]
                if blk.shape[1] and (int(blk[0].min()) < 0 or int(blk[2].min()) < 0 or int(blk[1].min()) < 0
                                     or int(blk[0].max()) >= N or int(blk[2].max()) >= N
                                     or int(blk[1].max()) > self.loader.num_kb_relation):
                    raise ValueError("question %d: node ids must lie in [0, %d), relation ids in [0, %d]"
                                     % (s_, N, self.loader.num_kb_relation))
                d = torch.from_numpy(np.ascontiguousarray(blk)).to(self.device)
                if len(self._dev) < self.max_questions:
                    self._dev[s_] = d
            parts.append(q)
            blocks.append(d)
        sizes = np.array([p[0].shape[1] for p in parts], dtype=np.int64)
        if blocks:
            hrt = _cat_blocks(blocks, sizes, N)
        else:
            hrt = torch.zeros((3, 0), dtype=torch.int32, device=self.device)
        return BatchFacts(hrt, sizes, parts, N, checked=True, checked_r1=int(self.loader.num_kb_relation) + 1, owner=self)


class DeviceStructureCache(DeviceFactCache):
    """:class:`DeviceFactCache` that also keeps every question's destination-sorted STRUCTURE on the GPU (a
    single-question ``ops.CsrPlan``, built the first time the question is used): a batch is then the
    concatenation of its questions' structures (``ops.CsrPlan.concat`` -> ``gnnrag_csr_concat``: copies with offsets, no
    upload, no sort, no wait for the stream) - SURVEY.md section 8 f-1 as written ("cached per-question int32 CSR built
    once at load time, batch = concatenation with offsets").

    Memory: a cached structure is the whole single-question layout - ~76 B per fact (record arrays of both directions,
    the merged stream, the id block) PLUS per-node arrays of the PADDED width (two row-pointer arrays, the big-node
    list; ~20 B per node slot, 256-byte aligned pieces): at N = 2000 that is ~50 KB per question before the first
    fact.  The cache is bounded by ``max_questions`` AND by a byte budget (``max_bytes``; default 16 GiB, environment
    ``GNNRAG_STRUCTURE_CACHE_MB``), least recently used questions are dropped first; a dropped question is simply
    rebuilt when it comes up again."""

    def __init__(self, loader, device, max_questions: int = 200000, max_bytes: int = None):
        import collections
        super().__init__(loader, device, max_questions)
        self._plans = collections.OrderedDict()
        if max_bytes is None:
            max_bytes = int(os.environ.get("GNNRAG_STRUCTURE_CACHE_MB", "16384")) << 20
        self.max_bytes = int(max_bytes)
        self.cached_bytes = 0

    @staticmethod
    def _plan_bytes(pl) -> int:
        return int(pl._mem.numel()) + int(pl._hrt.numel()) * 4

    def batch(self, sample_ids):
        import torch
        from .. import ops
        N = self.loader.max_local_entity
        R1 = self.loader.num_kb_relation + 1                  # rows of the relation feature tables (dataset_load.py:413)
        parts, plans = [], []
        for s_ in sample_ids:
            s_ = int(s_)
            q = self._question(s_)
            pl = self._plans.get(s_)
            if pl is None:
                blk = torch.from_numpy(np.ascontiguousarray(q[0])).to(self.device)
                pl = ops.CsrPlan(None, None, None, 1, N, R1, self.device, hrt_device=blk)
                self._plans[s_] = pl
                self.cached_bytes += self._plan_bytes(pl)
            else:
                self._plans.move_to_end(s_)
            parts.append(q)
            plans.append(pl)
        # bounds: questions of THIS batch stay alive through `plans` whatever is dropped from the cache
        while self._plans and (len(self._plans) > self.max_questions or self.cached_bytes > self.max_bytes):
            _, old = self._plans.popitem(last=False)
            self.cached_bytes -= self._plan_bytes(old)
        sizes = np.array([p[0].shape[1] for p in parts], dtype=np.int64)
        return BatchFacts(None, sizes, parts, N, plans=plans,
                          make_hrt=lambda: _cat_blocks([p._hrt[:, : p.F] for p in plans], sizes, N).to(self.device))


def question_fact_counts(loader, sample_ids) -> np.ndarray:
    """F_g of every question of a batch WITHOUT building its tuple: stored facts + self loops
    (``dataset_load.py:486,499-506``).  ``data_eff`` loaders build a question's facts on the fly - None (no cheap count)."""
    if loader.data_eff:
        return None
    n = np.array([len(loader.kb_adj_mats[int(s)][0]) for s in sample_ids], dtype=np.int64)
    if loader.use_self_loop:
        n += np.array([len(loader.global2local_entity_maps[int(s)]) for s in sample_ids], dtype=np.int64)
    return n


def patch_loader(loader, cache: bool = False, keep_rng_stream: bool = False, device=None, structures: bool = False,
                 shard=None):
    """Rebinds ``loader._build_fact_mat`` to the vectorised builder (the reference file is untouched).
    ``cache=True`` additionally serves ``fact_dropout == 0`` batches from a :class:`FactCache`.  The cached
    path does not draw the per-question ``np.random.permutation`` the reference draws even without dropout
    (``dataset_load.py:489-490``), so a run that interleaves cached evaluation batches with training batches
    (``train_model.py`` evaluates on ``valid`` every epoch) consumes a different numpy RNG stream than the
    reference; ``keep_rng_stream=True`` draws and discards those permutations (same stream as the reference,
    at the cost of most of the caching gain).  Use the plain cache for evaluation-only runs.  ``device``: keep the
    per-question id blocks on that GPU (:class:`DeviceFactCache`; the tuple is then a :class:`BatchFacts`, readable by
    the MI355X modules only).  ``structures`` (with ``device``): also cache every question's sorted structure on the GPU,
    so that a batch's structure is a concatenation (:class:`DeviceStructureCache`).  ``shard=(rank, world)`` (with ``cache``; evaluation under ``shard.shard_model``):
    a ``fact_dropout == 0`` batch of at least ``world`` questions is built for THIS RANK's fact-balanced question range
    only and handed out as :class:`ShardedFacts`."""
    fc = None
    if cache:
        if device is not None and structures:
            fc = DeviceStructureCache(loader, device)
        else:
            fc = DeviceFactCache(loader, device) if device is not None else FactCache(loader)

    def build(self, sample_ids, fact_dropout):
        if fc is not None and fact_dropout == 0:
            if keep_rng_stream:
                for sample_id in sample_ids:
                    n = len(self.create_kb_adj_mats(sample_id)[0]) if self.data_eff else len(self.kb_adj_mats[sample_id][0])
                    np.random.permutation(n)                                   # dataset_load.py:489
            if shard is not None and len(sample_ids) >= shard[1]:
                counts = question_fact_counts(self, sample_ids)
                if counts is not None:
                    from ..shard import balanced_ranges
                    ranges = balanced_ranges(counts, shard[1])
                    lo, hi = ranges[shard[0]]
                    return ShardedFacts(fc.batch(sample_ids[lo:hi]), counts, ranges, shard[0])
            return fc.batch(sample_ids)
        return build_fact_mat(self, sample_ids, fact_dropout)

    loader._build_fact_mat = types.MethodType(build, loader)
    return loader
