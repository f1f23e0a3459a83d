"""Tensor-level wrappers over the C ABI: device memory and streams come from
PyTorch-ROCm (plumbing), every computation is a call into libgnnrag_hip.so."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import _lib


MATH_FP32, MATH_BF16X3, MATH_MIXED = 0, 1, 2
MATH_NAMES = {MATH_FP32: "fp32 (v_mfma_f32_16x16x4_f32, bit-exact fmaf chains)",
              MATH_BF16X3: "bf16x3 (exact 3-way bf16 split, 6 plane products on v_mfma_f32_16x16x32_bf16, fp32 accumulate)",
              MATH_MIXED: "mixed (per kernel the faster fp32-class form: bf16x3 - exact 3-way bf16 split, 6 plane products, "
                          "fp32 accumulate - for the relation tables and the self-block update, exact fp32 MFMA for the "
                          "small products)"}
# Math mode the wrappers below pass to the library (the library itself keeps no mode: it is an argument of
# every dense entry point).  GNNRAG_MATH=fp32|bf16x3|mixed sets the binding's default.
_default_math = {"fp32": MATH_FP32, "bf16x3": MATH_BF16X3, "mixed": MATH_MIXED}[os.environ.get("GNNRAG_MATH", "mixed")]


def set_dense_math(mode: int) -> int:
    """Default math mode of this binding's dense calls: MATH_FP32 (exact fp32 MFMA), MATH_BF16X3 (exact 3-way bf16
    split, six plane products, fp32 accumulate) or MATH_MIXED (per kernel the faster of the two; the default).
    Returns the old mode."""
    global _default_math
    if mode not in (MATH_FP32, MATH_BF16X3, MATH_MIXED):
        raise ValueError("unknown math mode %r" % (mode,))
    old, _default_math = _default_math, int(mode)
    return old


def get_dense_math() -> int:
    return _default_math


def _math(math: Optional[int]) -> int:
    return _default_math if math is None else int(math)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=torch.float32, shape=None) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.GnnragError("%s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        t = t.contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


_stage_buf = {"t": None}


def _pinned_stage(n: int) -> torch.Tensor:
    """Pinned int32 staging block of at least n elements (grown geometrically, reused across batches)."""
    t = _stage_buf["t"]
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1 << 20, 0 if t is None else 2 * t.numel()), dtype=torch.int32, pin_memory=True)
        _stage_buf["t"] = t
    return t


def _on_plan_device(plan: "CsrPlan", t: torch.Tensor, name: str) -> None:
    """The structure's raw device pointers are only valid on the GPU it was built on."""
    if t.device != plan.device and not (t.is_cuda and plan.device.index is None):
        raise _lib.GnnragError("%s lives on %s but the structure was built on %s" % (name, t.device, plan.device))


class CsrPlan:
    """Device-side destination-sorted structure of one batch (both directions).

    Built from the first three arrays of the reference batch tuple
    (``kb_adj_mat`` = heads, rels, tails, ..., ``gnn/dataset_load.py:527``).
    Replaces ``BaseGNNLayer.build_matrix`` (``base_gnn.py:19-51``)."""

    def __init__(self, heads, rels, tails, B: int, N: int, R1: int, device, validate: bool = True,
                 hrt_device: Optional[torch.Tensor] = None, rel_counts=None):
        """``rel_counts = (rel_total, rel_max)``: the sum and the maximum over the questions of the distinct relation ids
        among a question's facts, when the caller knows them (a fact cache does: data/fact_mat.BatchFacts.rel_counts) -
        the build then does NOT wait for its stream (``gnnrag_csr_build_counts``); ``status()`` runs the deferred
        device-side validation whenever wanted."""
        lib = _lib.load()
        if hrt_device is not None:
            # the batch builder's [3, F] int32 block is already on the GPU (data/fact_mat.DeviceFactCache)
            if (hrt_device.dtype != torch.int32 or hrt_device.dim() != 2 or hrt_device.shape[0] != 3
                    or not hrt_device.is_cuda or not hrt_device.is_contiguous()):
                raise ValueError("hrt_device must be a contiguous [3, F] int32 CUDA tensor")
            heads = rels = tails = None
            F = int(hrt_device.shape[1])
        else:
            heads = np.asarray(heads)
            rels = np.asarray(rels)
            tails = np.asarray(tails)
            F = int(heads.shape[0])
        if hrt_device is None and (rels.shape[0] != F or tails.shape[0] != F):
            raise ValueError("heads/rels/tails differ in length")
        if B <= 0 or N <= 0 or R1 <= 0:
            raise ValueError("B, N, R1 must be positive")
        if B * N >= 2 ** 31 or F >= 2 ** 31:
            raise ValueError("batch too large for int32 indices")
        # ranges and the no-fact-across-questions rule are checked on the device during the build, on the
        # int32-narrowed ids (GNNRAG_E_TUPLE -> ValueError below); `validate` is kept for API compatibility
        self.B, self.N, self.R1, self.F = int(B), int(N), int(R1), F
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.GnnragError("CsrPlan needs a GPU device, got %s" % self.device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        base = None if hrt_device is not None else heads.base
        if hrt_device is not None:
            if hrt_device.device != self.device and not (self.device.index is None):
                raise _lib.GnnragError("hrt_device lives on %s, the plan on %s" % (hrt_device.device, self.device))
            hrt = hrt_device if F else torch.zeros((3, 1), dtype=torch.int32, device=hrt_device.device)
        elif (F and isinstance(base, np.ndarray) and heads.dtype == np.int32 and base is rels.base
                and base is tails.base and base.dtype == np.int32 and base.shape == (3, F) and base.flags.c_contiguous
                and heads.ctypes.data == base[0].ctypes.data and rels.ctypes.data == base[1].ctypes.data
                and tails.ctypes.data == base[2].ctypes.data and heads.shape == rels.shape == tails.shape == (F,)
                and heads.strides == rels.strides == tails.strides == (4,)):
            hrt = base                  # the batch builder's own [3,F] int32 block (data/fact_mat.py): no copy
        elif (F and heads.dtype == rels.dtype == tails.dtype == np.int64 and heads.flags.c_contiguous
              and rels.flags.c_contiguous and tails.flags.c_contiguous):
            # the reference's own tuple (int64 arrays): threaded narrowing into a pinned staging block, with the
            # check that every id fits int32 (ids that wrapped could pass the device-side range check)
            stage = _pinned_stage(3 * F)
            rc = lib.gnnrag_narrow_tuple(heads.ctypes.data, rels.ctypes.data, tails.ctypes.data, F, stage.data_ptr(),
                                         min(8, os.cpu_count() or 1))
            if rc == _lib.E_TUPLE:
                raise ValueError("edge tuple out of range: ids must lie in [0, 2^31)")
            _lib.check(rc, "gnnrag_narrow_tuple")
            hrt = stage[: 3 * F].view(3, F)
        else:
            for name, a in (("heads", heads), ("rels", rels), ("tails", tails)):
                if F and a.dtype.itemsize > 4 and (int(a.max()) >= 2 ** 31 or int(a.min()) < 0):
                    raise ValueError("edge tuple out of range: %s holds ids outside [0, 2^31)" % name)
            hrt = np.empty((3, max(F, 1)), dtype=np.int32)
            hrt[0, :F], hrt[1, :F], hrt[2, :F] = heads, rels, tails
        with torch.cuda.device(self.device):
            if hrt_device is not None:
                self._hrt = hrt                                                         # already resident
            elif isinstance(hrt, np.ndarray):
                self._hrt = torch.from_numpy(hrt).to(self.device, non_blocking=False)   # ONE int32 upload
            else:
                self._hrt = hrt.to(self.device, non_blocking=True)                      # from the pinned block
                torch.cuda.current_stream().synchronize()                               # the block is reused next batch
            nbytes = lib.gnnrag_csr_bytes(F, B, N, R1, 0, 0)
            sbytes = lib.gnnrag_csr_scratch_bytes(F, B, N, R1)
            self._mem = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            scratch = torch.empty(max(sbytes, 256), dtype=torch.uint8, device=self.device)
            self.c = _lib.CsrStruct()
            row = self._hrt
            rt, rm = (-1, -1) if rel_counts is None else (int(rel_counts[0]), int(rel_counts[1]))
            rc = lib.gnnrag_csr_build_counts(
                row[0].data_ptr(), row[1].data_ptr(), row[2].data_ptr(), None, None,
                F, B, N, R1, rt, rm, self._mem.data_ptr(), self._mem.numel(),
                scratch.data_ptr(), scratch.numel(), C.byref(self.c), _stream())
            if rel_counts is not None:
                scratch.record_stream(torch.cuda.current_stream())      # the build has not run yet: keep its scratch
            if rc == _lib.E_TUPLE:
                raise ValueError("edge tuple out of range: node ids must lie in [0, B*N), relation ids in [0, R1), "
                                 "and a fact may not connect two different questions")
            _lib.check(rc, "gnnrag_csr_build")
            # the build waits for its stream once (it returns the relation counts), so scratch is free
        self._w = {}
        # compact relation rows: question b's tables are rows rel_off[b] : rel_off[b+1] of P[d]
        self.rel_total, self.rel_max = int(self.c.rel_total), int(self.c.rel_max)

    def status(self) -> None:
        """The deferred check of a structure built with ``rel_counts`` (waits for the stream once): raises ``ValueError``
        for an invalid tuple, ``GnnragError`` when the counts passed in differ from what the device counted."""
        rc = _lib.load().gnnrag_csr_status(C.byref(self.c), _stream())
        if rc == _lib.E_TUPLE:
            raise ValueError("edge tuple out of range: node ids must lie in [0, B*N), relation ids in [0, R1), "
                             "and a fact may not connect two different questions")
        _lib.check(rc, "gnnrag_csr_status (relation counts passed to the build differ from the device's)")

    @classmethod
    def concat(cls, parts, N: int, R1: int, device) -> "CsrPlan":
        """The structure of a batch as the concatenation of per-question structures already on the device
        (``gnnrag_csr_concat``; ``parts`` = one ``CsrPlan(..., B=1, N, R1)`` per question, in batch order): a copy with
        offsets - no upload, no sort, no wait for the stream; bit-identical to building the batch tuple from scratch."""
        lib = _lib.load()
        B = len(parts)
        if B == 0:
            raise ValueError("a batch needs at least one question")
        self = cls.__new__(cls)
        self.B, self.N, self.R1 = B, int(N), int(R1)
        self.F = int(sum(p.F for p in parts))
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        for p in parts:
            if p.B != 1 or p.N != self.N or p.R1 != self.R1 or p.device != self.device:
                raise ValueError("parts must be single-question structures of the same N / R1 on %s" % self.device)
        if B * self.N >= 2 ** 31 or self.F >= 2 ** 31:
            raise ValueError("batch too large for int32 indices")
        arr = (C.POINTER(_lib.CsrStruct) * B)(*[C.pointer(p.c) for p in parts])
        with torch.cuda.device(self.device):
            nbytes = lib.gnnrag_csr_bytes(self.F, B, self.N, self.R1, 0, 0)
            self._mem = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            self.c = _lib.CsrStruct()
            _lib.check(lib.gnnrag_csr_concat(arr, B, self.N, self.R1, self._mem.data_ptr(), self._mem.numel(),
                                             C.byref(self.c), _stream()), "gnnrag_csr_concat")
        self._parts = list(parts)            # the per-question blocks stay alive as long as the batch may read them
        self._hrt_lazy = None
        self._w = {}
        self.rel_total, self.rel_max = int(self.c.rel_total), int(self.c.rel_max)
        return self

    @property
    def _hrt(self):
        """[3, F] int32 id block of the batch tuple on the device (the backward's (question, relation) ordering reads
        it).  A concatenated structure builds it from its parts on first use."""
        if getattr(self, "_hrt_lazy", None) is None:
            blocks = []
            for b, p in enumerate(self._parts):
                h = p._hrt[:, : p.F].clone()
                h[0] += b * self.N
                h[2] += b * self.N
                blocks.append(h)
            self._hrt_lazy = torch.cat(blocks, dim=1) if self.F else torch.zeros((3, 1), dtype=torch.int32, device=self.device)
        return self._hrt_lazy

    @_hrt.setter
    def _hrt(self, value):
        self._hrt_lazy = value

    def walk_workspace(self, D: int, I: int) -> torch.Tensor:
        """Scratch for the heavy-row partial sums of the walk kernels (cached per (D, I))."""
        key = ("ws", D, min(I, 3))
        if key not in self._w:
            nbytes = _lib.load().gnnrag_aggregate_workspace_bytes(C.byref(self.c), D, I)
            self._w[key] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self._w[key]

    def relorder(self):
        """Facts ordered by (question, relation) for the backward's gather kernels; built on first use
        (training only).  Returns the ctypes struct (device arrays are owned by the plan)."""
        if ("relorder",) not in self._w:
            lib = _lib.load()
            w = self._w.get(("w_gnn_src",))
            with torch.cuda.device(self.device):
                mem = torch.empty(max(lib.gnnrag_relorder_bytes(C.byref(self.c), int(w is not None)), 256),
                                  dtype=torch.uint8, device=self.device)
                scratch = torch.empty(max(lib.gnnrag_relorder_scratch_bytes(C.byref(self.c)), 256),
                                      dtype=torch.uint8, device=self.device)
                ro = _lib.RelorderStruct()
                row = self._hrt
                _lib.check(lib.gnnrag_relorder_build(
                    C.byref(self.c), row[0].data_ptr(), row[1].data_ptr(), row[2].data_ptr(), _ptr(w),
                    mem.data_ptr(), mem.numel(), scratch.data_ptr(), scratch.numel(), C.byref(ro), _stream()),
                    "gnnrag_relorder_build")
            self._w[("relorder",)] = (ro, mem)
        return self._w[("relorder",)][0]

    def backward_workspace(self, D: int, I: int, ro=None) -> torch.Tensor:
        """Scratch of the backward kernels, cached per (D, I)."""
        key = ("bws", D, I, ro is not None)
        if key not in self._w:
            nbytes = _lib.load().gnnrag_backward_workspace_bytes(C.byref(self.c), None if ro is None else C.byref(ro),
                                                                 D, I)
            self._w[key] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self._w[key]

    # -- lazily attached per-fact weights ----------------------------------------------------
    def _attach(self, key: str, w_per_fact, square: bool):
        if key in self._w:
            return
        lib = _lib.load()
        w = np.asarray(w_per_fact, dtype=np.float32)
        if w.shape[0] != self.F:
            raise ValueError("%s has %d entries for %d facts" % (key, w.shape[0], self.F))
        with torch.cuda.device(self.device):
            src = torch.from_numpy(w).to(self.device) if self.F else torch.zeros(1, device=self.device)
            out = torch.empty((2, max(self.F, 1)), dtype=torch.float32, device=self.device)
            _lib.check(lib.gnnrag_csr_permute_weight(C.byref(self.c), src.data_ptr(), int(square),
                                                     out[0].data_ptr(), out[1].data_ptr(), _stream()),
                       "gnnrag_csr_permute_weight")
        self._w[key] = out
        if key == "w_gnn":
            if ("relorder",) in self._w:
                raise RuntimeError("attach_w_gnn after the backward structure was built")
            self._w[("w_gnn_src",)] = src            # original fact order: the (question, relation) ordering permutes it too
        if key == "w_rel":
            self._w[("w_rel_src",)] = src            # read through relorder.perm by the TypeLayer backward
        arr = getattr(self.c, key)
        arr[0], arr[1] = out[0].data_ptr(), out[1].data_ptr()

    def attach_w_gnn(self, weight_list):
        """``weight_list`` = 1/outdeg(head); used squared when ``normalized_gnn`` (base_gnn.py:38-47)."""
        self._attach("w_gnn", weight_list, True)

    def attach_w_rel(self, weight_rel_list):
        """``weight_rel_list`` = 1/count(head, rel); TypeLayer ``norm_rel`` (layer_init.py:39-40)."""
        self._attach("w_rel", weight_rel_list, False)

    # -- debug / test views --------------------------------------------------------------------
    def rel_rows_device(self) -> torch.Tensor:
        """[rel_total, 2] int32 (question, relation id) of every compact relation row, a view of the structure's own
        device memory (the differentiable relation tables of training index T and the instructions with it)."""
        if self.rel_total == 0:
            return torch.zeros((0, 2), dtype=torch.int32, device=self.device)
        off = int(C.cast(self.c.rel_rows, C.c_void_p).value) - self._mem.data_ptr()
        return self._mem[off: off + 8 * self.rel_total].view(torch.int32).view(-1, 2)

    def _view(self, addr: int, n: int, dtype):
        if n == 0:
            return torch.zeros(0, dtype=dtype)
        base = self._mem.data_ptr()
        off = addr - base
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        return self._mem[off: off + nbytes].view(dtype).cpu()

    def to_host(self) -> dict:
        """Copies the structure back (tests only)."""
        BN = self.B * self.N
        out = {}
        for d in (0, 1):
            out["row_ptr%d" % d] = self._view(self.c.row_ptr[d], BN + 1, torch.int32).numpy()
            out["edge%d" % d] = self._view(self.c.edge[d], 2 * self.F, torch.int32).numpy().reshape(-1, 2)
            out["perm%d" % d] = self._view(self.c.perm[d], self.F, torch.int32).numpy()
        nh = self._view(self.c.n_heavy, 2, torch.int32).numpy()
        out["n_heavy"] = nh
        for d in (0, 1):
            out["heavy%d" % d] = self._view(self.c.heavy[d], int(min(nh[d], self.c.heavy_cap)), torch.int32).numpy()
            out["hub_q_off%d" % d] = self._view(self.c.hub_q_off[d], self.B + 1, torch.int32).numpy()
            out["hub_wbase%d" % d] = self._view(self.c.hub_wbase[d], self.B + 1, torch.int32).numpy()
        for key, t in self._w.items():
            if isinstance(key, str):
                out[key] = t.cpu().numpy()
        nc = self._view(self.c.n_chunks, 2, torch.int32).numpy()
        out["n_chunks"] = nc
        bc = self._view(self.c.big_cnt, self.B, torch.int32).numpy()
        bn = self._view(self.c.big_nodes, BN, torch.int32).numpy().reshape(self.B, self.N)
        out["big"] = [np.sort(bn[b, : bc[b]]) for b in range(self.B)]
        for d in (0, 1):
            out["edge_l%d" % d] = self._view(self.c.edge_l[d], 2 * self.F, torch.int32).numpy().reshape(-1, 2)
        out["rel_off"] = self._view(self.c.rel_off, self.B + 1, torch.int32).numpy()
        out["rel_rows"] = self._view(self.c.rel_rows, 2 * self.rel_total, torch.int32).numpy().reshape(-1, 2)
        out["edge_m"] = self._view(self.c.edge_m, 4 * self.F, torch.int32).numpy().reshape(-1, 2)
        out["m_from"] = self._view(self.c.m_from, 2 * self.F, torch.int32).numpy()
        out["m_dst"] = self._view(self.c.m_dst, 2 * self.F, torch.int32).numpy()
        return out

    def rel_rows(self) -> np.ndarray:
        """[rel_total, 2] (question, relation id) of every compact relation row (host copy)."""
        return self._view(self.c.rel_rows, 2 * self.rel_total, torch.int32).numpy().reshape(-1, 2)


def linear(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
           add: Optional[torch.Tensor] = None, relu: bool = False, math: Optional[int] = None) -> torch.Tensor:
    """act(A W^T + bias (+ add on the first add.shape[0] rows)) on fp32 MFMA."""
    lib = _lib.load()
    A = _chk(A, "A")
    W = _chk(W, "W")
    M, K = A.shape
    Nout = W.shape[0]
    if W.shape[1] != K:
        raise ValueError("W is %s, A is %s" % (tuple(W.shape), tuple(A.shape)))
    bias = None if bias is None else _chk(bias, "bias", shape=(Nout,))
    add_rows = 0
    if add is not None:
        add = _chk(add, "add")
        if add.shape[1] != Nout:
            raise ValueError("add must have %d columns" % Nout)
        add_rows = add.shape[0]
    out = torch.empty((M, Nout), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        _lib.check(lib.gnnrag_linear(A.data_ptr(), M, K, W.data_ptr(), _ptr(bias), _ptr(add), add_rows,
                                     int(relu), out.data_ptr(), Nout, _math(math), _stream()), "gnnrag_linear")
    return out


def rel_transform(relfeat: torch.Tensor, relfeat_inv: torch.Tensor, layers, planes: bool = False):
    """Relation projections of all layers in one launch (reasongnn.py:75-79, :102-105):
    ``out[j, d] = rel_linear{j}(rel_features_d) (+ pos_emb{j}_d on its rows)``, d = 0 forward / 1 inverse.
    ``layers``: sequence of (W_rel [D,D], b_rel [D], pos_emb.weight or None, pos_emb_inv.weight or None).
    Returns [L, 2, R1, D]; with ``planes=True`` also the bf16 planes of relu(+-T) ([L, 2, 3, R1, 448] int16 view) that
    :func:`relation_tables_planes` multiplies."""
    lib = _lib.load()
    relfeat = _chk(relfeat, "rel_features")
    R1, D = relfeat.shape
    relfeat_inv = _chk(relfeat_inv, "rel_features_inv", shape=(R1, D))
    L = len(layers)
    params = (_lib.LayerParams * max(L, 1))()
    keep, pos_rows = [], 0
    for j, (W, b, pos, pos_inv) in enumerate(layers):
        W = _chk(W, "rel_linear.weight", shape=(D, D))
        b = _chk(b, "rel_linear.bias", shape=(D,))
        keep += [W, b]
        params[j].W_rel, params[j].b_rel = W.data_ptr(), b.data_ptr()
        if pos is not None:
            pos = _chk(pos, "pos_emb.weight")
            pos_inv = _chk(pos_inv, "pos_emb_inv.weight", shape=tuple(pos.shape))
            if pos.shape[1] != D or (pos_rows and pos.shape[0] != pos_rows):
                raise ValueError("pos_emb must be [rows, D], the same size in every layer")
            pos_rows = pos.shape[0]
            keep += [pos, pos_inv]
            params[j].pos_fwd, params[j].pos_inv = pos.data_ptr(), pos_inv.data_ptr()
    out = torch.empty((L, 2, R1, D), dtype=torch.float32, device=relfeat.device)
    pl = None
    if planes:
        nbytes = lib.gnnrag_rel_planes_bytes(R1, D, L)
        if nbytes == 0:
            raise _lib.GnnragError("relation planes need a hidden size <= 224")
        pl = torch.empty((L, 2, 3, R1, nbytes // (L * 6 * R1 * 2)), dtype=torch.int16, device=relfeat.device)
    with torch.cuda.device(relfeat.device):
        _lib.check(lib.gnnrag_rel_transform(relfeat.data_ptr(), relfeat_inv.data_ptr(), R1, D, L, params, pos_rows,
                                            out.data_ptr(), _ptr(pl), _stream()), "gnnrag_rel_transform")
    return (out, pl) if planes else out


def relation_tables_planes(plan: "CsrPlan", planes: torch.Tensor, ins: torch.Tensor, W_e2e: torch.Tensor) -> torch.Tensor:
    """Relation tables of ONE layer in the bf16x3 math mode from its pre-split relation planes (``rel_transform(...,
    planes=True)[1][j]``, [2, 3, R1, 448] int16): ``gnnrag_relation_tables_planes``.  Raises outside the kernel's
    shapes (193 <= D <= 208, rel_total >= 1024)."""
    lib = _lib.load()
    ins = _chk(ins, "relational_ins")
    B, I, D = ins.shape
    if (planes.dtype != torch.int16 or tuple(planes.shape[:3]) != (2, 3, plan.R1) or not planes.is_contiguous()
            or B != plan.B):
        raise ValueError("planes must be a contiguous [2, 3, R1, 448] int16 tensor of this plan's relation count")
    W_e2e = _chk(W_e2e, "e2e_linear.weight", shape=(D, (2 * I + 1) * D))
    _on_plan_device(plan, ins, "relational_ins")
    P = torch.empty((2, max(plan.rel_total, 1), D), dtype=torch.float32, device=ins.device)
    with torch.cuda.device(ins.device):
        _lib.check(lib.gnnrag_relation_tables_planes(C.byref(plan.c), planes.data_ptr(), ins.data_ptr(), W_e2e.data_ptr(),
                                                     P.data_ptr(), D, I, _stream()), "gnnrag_relation_tables_planes")
    return P[:, :plan.rel_total]


def gemm_tn(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """``A^T @ B`` for two row-major operands with the same (large) row count - the weight gradient of a dense
    projection, ``dW = dY^T X`` (``gnnrag_gemm_tn``: exact fp32 MFMA, fixed summation order)."""
    lib = _lib.load()
    A = _chk(A, "A")
    B = _chk(B, "B")
    if A.dim() != 2 or B.dim() != 2 or A.shape[0] != B.shape[0]:
        raise ValueError("A is %s, B is %s" % (tuple(A.shape), tuple(B.shape)))
    M, N1 = A.shape
    N2 = B.shape[1]
    out = torch.empty((N1, N2), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        # the chunk count behind the workspace size depends on the CURRENT device's CU count: query it on A's device
        ws = torch.empty(max(lib.gnnrag_gemm_tn_workspace_bytes(M, N1, N2), 16), dtype=torch.uint8, device=A.device)
        _lib.check(lib.gnnrag_gemm_tn(A.data_ptr(), B.data_ptr(), M, N1, N2, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _stream()), "gnnrag_gemm_tn")
    return out


def aggregate(plan: CsrPlan, dist: torch.Tensor, ins: torch.Tensor, T_fwd: torch.Tensor,
              T_inv: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    B, N = plan.B, plan.N
    ins = _chk(ins, "ins")
    _, I, D = ins.shape
    dist = _chk(dist, "dist").reshape(-1)
    _on_plan_device(plan, dist, "dist")
    if dist.numel() != B * N or ins.shape[0] != B:
        raise ValueError("dist/ins do not match the plan (B=%d, N=%d)" % (B, N))
    T_fwd = _chk(T_fwd, "T_fwd", shape=(plan.R1, D))
    T_inv = _chk(T_inv, "T_inv", shape=(plan.R1, D))
    agg = torch.empty((B * N, 2 * I * D), dtype=torch.float32, device=dist.device)
    ws = plan.walk_workspace(D, I)
    with torch.cuda.device(dist.device):
        _lib.check(lib.gnnrag_aggregate(C.byref(plan.c), dist.data_ptr(), ins.data_ptr(), T_fwd.data_ptr(),
                                        T_inv.data_ptr(), agg.data_ptr(), D, I, ws.data_ptr(), ws.numel(),
                                        _stream()), "gnnrag_aggregate")
    return agg


def relation_tables(plan: CsrPlan, T_fwd: torch.Tensor, T_inv: torch.Tensor, ins: torch.Tensor,
                    W_e2e: torch.Tensor, math: Optional[int] = None) -> torch.Tensor:
    """P[d,row(b,r),:] = sum_i W_e2e[:, block(i,d)] relu(T_d[r,:] * ins[b,i,:])  ->  [2,rel_total,D],
    one row per (question, relation the question uses) - ``plan.rel_rows()`` lists them."""
    lib = _lib.load()
    ins = _chk(ins, "ins")
    B, I, D = ins.shape
    if B != plan.B:
        raise ValueError("ins has %d questions, the plan %d" % (B, plan.B))
    T_fwd = _chk(T_fwd, "T_fwd", shape=(plan.R1, D))
    T_inv = _chk(T_inv, "T_inv", shape=(plan.R1, D))
    W_e2e = _chk(W_e2e, "e2e_linear.weight", shape=(D, (2 * I + 1) * D))
    P = torch.empty((2, plan.rel_total, D), dtype=torch.float32, device=ins.device)
    with torch.cuda.device(ins.device):
        _lib.check(lib.gnnrag_relation_tables(C.byref(plan.c), T_fwd.data_ptr(), T_inv.data_ptr(), ins.data_ptr(),
                                              W_e2e.data_ptr(), P.data_ptr(), D, I, _math(math), _stream()),
                   "gnnrag_relation_tables")
    return P


WALK_L2_GATHER, WALK_LDS_16, WALK_LDS_32 = 0, 1, 2
WALK_KERNEL_NAMES = {WALK_L2_GATHER: "k_walk_light_q / k_walk_light<FUSED> + hub kernels (table rows gathered from L2)",
                     WALK_LDS_16: "k_fact_prior_merged + k_walk_slice<FUSED,1> (16-column table slices in LDS, merged rows)",
                     WALK_LDS_32: "k_fact_prior_merged + k_walk_slice<FUSED,2> (32-column table slices in LDS, merged rows)"}


def aggregate_fused_variant(plan: CsrPlan, D: int) -> int:
    """Which kernel ``aggregate_fused`` runs for this structure and hidden size (WALK_*)."""
    v = _lib.load().gnnrag_aggregate_fused_variant(C.byref(plan.c), int(D))
    if v < 0:
        _lib.check(v, "gnnrag_aggregate_fused_variant")
    return v


HUB_FORM_NONE, HUB_FORM_DENSE, HUB_FORM_CHUNKED = 0, 1, 2


def aggregate_fused_hub_form(plan: CsrPlan, D: int, I: int = 1) -> dict:
    """What the hub rows of a fused aggregation call do for this structure: the device-side decision of the gather walk
    (dense product or chunked fallback), read back.  ``I`` sizes the workspace as the layer call does
    (``gnnrag_aggregate_workspace_bytes(csr, D, I)``, softmax_layer.hip layer_ws)."""
    lib = _lib.load()
    ws = plan.walk_workspace(D, I)
    form = torch.zeros(4, dtype=torch.int32, device=ws.device)
    with torch.cuda.device(ws.device):
        _lib.check(lib.gnnrag_aggregate_fused_hub_form(C.byref(plan.c), int(D), ws.data_ptr(), ws.numel(), form.data_ptr(),
                                                       _stream()), "gnnrag_aggregate_fused_hub_form")
    f = form.cpu().tolist()
    return {"form": f[0], "hubs": (f[1], f[2]), "relation_ranges": f[3]}


def aggregate_fused(plan: CsrPlan, dist: torch.Tensor, P: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    B, N = plan.B, plan.N
    P = _chk(P, "P")
    D = P.shape[-1]
    if tuple(P.shape) != (2, plan.rel_total, D):
        raise ValueError("P must be [2, plan.rel_total, D]")
    dist = _chk(dist, "dist").reshape(-1)
    _on_plan_device(plan, dist, "dist")
    out = torch.empty((B * N, D), dtype=torch.float32, device=dist.device)
    ws = plan.walk_workspace(D, 1)
    with torch.cuda.device(dist.device):
        _lib.check(lib.gnnrag_aggregate_fused(C.byref(plan.c), dist.data_ptr(), P.data_ptr(), out.data_ptr(), D,
                                              ws.data_ptr(), ws.numel(), _stream()), "gnnrag_aggregate_fused")
    return out


class Frontier:
    """The frontier of a sparse prior (``gnnrag_frontier_build``): the nodes reached by the facts that start at a node
    with ``dist != 0`` and the compact relation rows those facts use.  Test-side view of what
    ``GNNRAG_PATH_SEED_PRIOR`` does inside ``gnnrag_reason_layer`` / ``gnnrag_reason_stack``."""

    def __init__(self, plan: CsrPlan, dist: torch.Tensor):
        lib = _lib.load()
        self.plan = plan
        dist = _chk(dist, "dist").reshape(-1)
        _on_plan_device(plan, dist, "dist")
        if dist.numel() != plan.B * plan.N:
            raise ValueError("dist does not match the plan")
        self.dist = dist
        nbytes = max(lib.gnnrag_frontier_workspace_bytes(C.byref(plan.c)), 256)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dist.device)
        with torch.cuda.device(dist.device):
            _lib.check(lib.gnnrag_frontier_build(C.byref(plan.c), dist.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                                 _stream()), "gnnrag_frontier_build")

    def read(self):
        """(number of listed nodes, number of listed relation rows, row gates uint8 [B*N]) - synchronises."""
        counts = (C.c_int32 * 2)()
        flags = np.zeros(self.plan.B * self.plan.N, dtype=np.uint8)
        with torch.cuda.device(self.dist.device):
            _lib.check(_lib.load().gnnrag_frontier_read(C.byref(self.plan.c), self.ws.data_ptr(), counts,
                                                        flags.ctypes.data, _stream()), "gnnrag_frontier_read")
        return int(counts[0]), int(counts[1]), flags

    def relation_tables(self, T_fwd, T_inv, ins, W_e2e, P: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The listed rows of P [2, rel_total, D] (the rest keeps what ``P`` held; NaN-filled when allocated here)."""
        ins = _chk(ins, "ins")
        B, I, D = ins.shape
        T_fwd = _chk(T_fwd, "T_fwd", shape=(self.plan.R1, D))
        T_inv = _chk(T_inv, "T_inv", shape=(self.plan.R1, D))
        W_e2e = _chk(W_e2e, "e2e_linear.weight", shape=(D, (2 * I + 1) * D))
        if P is None:
            P = torch.full((2, max(self.plan.rel_total, 1), D), float("nan"), dtype=torch.float32, device=ins.device)
        with torch.cuda.device(ins.device):
            _lib.check(_lib.load().gnnrag_relation_tables_frontier(
                C.byref(self.plan.c), self.ws.data_ptr(), T_fwd.data_ptr(), T_inv.data_ptr(), ins.data_ptr(),
                W_e2e.data_ptr(), P.data_ptr(), D, I, _stream()), "gnnrag_relation_tables_frontier")
        return P

    def aggregate(self, P: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """nbr rows of the listed nodes ([B*N, D]; unlisted rows keep what ``out`` held, zeros when allocated here)."""
        P = _chk(P, "P")
        D = P.shape[-1]
        if out is None:
            out = torch.zeros((self.plan.B * self.plan.N, D), dtype=torch.float32, device=P.device)
        with torch.cuda.device(P.device):
            _lib.check(_lib.load().gnnrag_aggregate_fused_frontier(
                C.byref(self.plan.c), self.ws.data_ptr(), self.dist.data_ptr(), P.data_ptr(), out.data_ptr(), D,
                _stream()), "gnnrag_aggregate_fused_frontier")
        return out


def update_score_fused(h, nbr, W, b, w_s, b_s, mask, I: int, math: Optional[int] = None):
    lib = _lib.load()
    h = _chk(h, "h")
    BN, D = h.shape
    nbr = _chk(nbr, "nbr", shape=(BN, D))
    W = _chk(W, "W", shape=(D, (2 * I + 1) * D))
    b = _chk(b, "b", shape=(D,))
    w_s = _chk(w_s, "w_s").reshape(-1)
    b_s = _chk(b_s, "b_s").reshape(-1)
    mask = _chk(mask, "mask").reshape(-1)
    h_out = torch.empty_like(h)
    score = torch.empty(BN, dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        _lib.check(lib.gnnrag_update_score_fused(h.data_ptr(), nbr.data_ptr(), W.data_ptr(), b.data_ptr(),
                                                 w_s.data_ptr(), b_s.data_ptr(), mask.data_ptr(), h_out.data_ptr(),
                                                 score.data_ptr(), BN, D, I, _math(math), _stream()),
                   "gnnrag_update_score_fused")
    return h_out, score


def update_score(h, agg, W, b, w_s, b_s, mask, I: int, math: Optional[int] = None):
    lib = _lib.load()
    h = _chk(h, "h")
    BN, D = h.shape
    agg = _chk(agg, "agg", shape=(BN, 2 * I * D))
    W = _chk(W, "W", shape=(D, (2 * I + 1) * D))
    b = _chk(b, "b", shape=(D,))
    w_s = _chk(w_s, "w_s").reshape(-1)
    b_s = _chk(b_s, "b_s").reshape(-1)
    mask = _chk(mask, "mask").reshape(-1)
    if w_s.numel() != D or b_s.numel() != 1 or mask.numel() != BN:
        raise ValueError("score_func / mask shapes do not match")
    h_out = torch.empty_like(h)
    score = torch.empty(BN, dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        _lib.check(lib.gnnrag_update_score(h.data_ptr(), agg.data_ptr(), W.data_ptr(), b.data_ptr(),
                                           w_s.data_ptr(), b_s.data_ptr(), mask.data_ptr(), h_out.data_ptr(),
                                           score.data_ptr(), BN, D, I, _math(math), _stream()), "gnnrag_update_score")
    return h_out, score


def masked_softmax(score: torch.Tensor, B: int, N: int) -> torch.Tensor:
    lib = _lib.load()
    score = _chk(score, "score")
    if score.numel() != B * N:
        raise ValueError("score has %d entries, expected %d" % (score.numel(), B * N))
    dist = torch.empty((B, N), dtype=torch.float32, device=score.device)
    with torch.cuda.device(score.device):
        _lib.check(lib.gnnrag_masked_softmax(score.data_ptr(), dist.data_ptr(), B, N, _stream()),
                   "gnnrag_masked_softmax")
    return dist


def typelayer(plan: CsrPlan, T: torch.Tensor, use_w_rel: bool) -> torch.Tensor:
    lib = _lib.load()
    T = _chk(T, "T")
    _on_plan_device(plan, T, "T")
    D = T.shape[1]
    if T.shape[0] != plan.R1:
        raise ValueError("T has %d rows, plan has R1=%d" % (T.shape[0], plan.R1))
    h0 = torch.empty((plan.B * plan.N, D), dtype=torch.float32, device=T.device)
    ws = plan.walk_workspace(D, 1)
    with torch.cuda.device(T.device):
        _lib.check(lib.gnnrag_typelayer(C.byref(plan.c), T.data_ptr(), int(use_w_rel), h0.data_ptr(), D,
                                        ws.data_ptr(), ws.numel(), _stream()), "gnnrag_typelayer")
    return h0


def aggregate_backward(plan: CsrPlan, dist, ins, T_fwd, T_inv, g_agg, gather: bool = True):
    """Gradients of ``aggregate`` with respect to (dist, ins, T_fwd, T_inv).  ``gather``: table / instruction
    gradients by the atomic-free gather over (question, relation) rows (needs D % 4 == 0, I <= 4; builds the
    ordering on first use) instead of relation-bucketed LDS sums."""
    lib = _lib.load()
    B, N = plan.B, plan.N
    ins = _chk(ins, "ins")
    _, I, D = ins.shape
    dist = _chk(dist, "dist").reshape(-1)
    T_fwd = _chk(T_fwd, "T_fwd", shape=(plan.R1, D))
    T_inv = _chk(T_inv, "T_inv", shape=(plan.R1, D))
    g_agg = _chk(g_agg, "g_agg", shape=(B * N, 2 * I * D))
    g_dist = torch.empty(B * N, dtype=torch.float32, device=dist.device)
    g_ins = torch.empty_like(ins)
    g_Tf = torch.empty_like(T_fwd)
    g_Ti = torch.empty_like(T_inv)
    ro = plan.relorder() if (gather and D % 4 == 0 and I <= 4) else None
    ws = plan.backward_workspace(D, I, ro)
    with torch.cuda.device(dist.device):
        _lib.check(lib.gnnrag_aggregate_backward(
            C.byref(plan.c), None if ro is None else C.byref(ro), dist.data_ptr(), ins.data_ptr(), T_fwd.data_ptr(),
            T_inv.data_ptr(), g_agg.data_ptr(),
            g_dist.data_ptr(), g_ins.data_ptr(), g_Tf.data_ptr(), g_Ti.data_ptr(), D, I, ws.data_ptr(), ws.numel(),
            _stream()), "gnnrag_aggregate_backward")
    return g_dist, g_ins, g_Tf, g_Ti


def aggregate_fused_backward(plan: CsrPlan, dist, P, g_nbr):
    """Gradients of ``aggregate_fused`` with respect to (dist, P): g_dist [BN], g_P [2, rel_total, D]."""
    lib = _lib.load()
    P = _chk(P, "P")
    D = P.shape[-1]
    if tuple(P.shape) != (2, plan.rel_total, D) or D % 4:
        raise ValueError("P must be [2, plan.rel_total, D] with D % 4 == 0")
    dist = _chk(dist, "dist").reshape(-1)
    g_nbr = _chk(g_nbr, "g_nbr", shape=(plan.B * plan.N, D))
    g_dist = torch.empty(plan.B * plan.N, dtype=torch.float32, device=dist.device)
    g_P = torch.zeros_like(P) if plan.rel_total == 0 else torch.empty_like(P)
    ro = plan.relorder()
    ws = plan.backward_workspace(D, 1, ro)
    with torch.cuda.device(dist.device):
        _lib.check(lib.gnnrag_aggregate_fused_backward(
            C.byref(plan.c), C.byref(ro), dist.data_ptr(), P.data_ptr(), g_nbr.data_ptr(), g_dist.data_ptr(),
            g_P.data_ptr(), D, ws.data_ptr(), ws.numel(), _stream()), "gnnrag_aggregate_fused_backward")
    return g_dist, g_P


def typelayer_backward(plan: CsrPlan, g_pre: torch.Tensor, use_w_rel: bool, gather: bool = True) -> torch.Tensor:
    """Gradient of ``typelayer`` with respect to T; g_pre = gradient of the pre-activation [BN, D].
    ``gather``: atomic-free gather over (question, relation) rows (D % 4 == 0) instead of LDS sums."""
    lib = _lib.load()
    g_pre = _chk(g_pre, "g_pre")
    D = g_pre.shape[1]
    if g_pre.shape[0] != plan.B * plan.N:
        raise ValueError("g_pre has %d rows, the plan %d nodes" % (g_pre.shape[0], plan.B * plan.N))
    g_T = torch.empty((plan.R1, D), dtype=torch.float32, device=g_pre.device)
    ro = plan.relorder() if (gather and D % 4 == 0) else None
    w_src = plan._w.get(("w_rel_src",)) if use_w_rel else None
    if use_w_rel and w_src is None:
        raise ValueError("use_w_rel needs attach_w_rel first")
    ws = plan.backward_workspace(D, 1, ro)
    with torch.cuda.device(g_pre.device):
        _lib.check(lib.gnnrag_typelayer_backward(C.byref(plan.c), None if ro is None else C.byref(ro), g_pre.data_ptr(),
                                                 _ptr(w_src), int(use_w_rel), g_T.data_ptr(), D, ws.data_ptr(),
                                                 ws.numel(), _stream()), "gnnrag_typelayer_backward")
    return g_T


class LayerWorkspace:
    """Scratch of gnnrag_reason_layer (T_fwd, T_inv, agg), reused across layer calls."""

    def __init__(self):
        self.key = None
        self.buf = None

    def get(self, plan: "CsrPlan", D, I, device) -> torch.Tensor:
        nbytes = max(_lib.load().gnnrag_layer_workspace_bytes(C.byref(plan.c), D, I), 256)
        if self.key != str(device) or self.buf is None or self.buf.numel() < nbytes:
            self.buf = None                      # release the old buffer before taking the new one
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self.key = str(device)
        return self.buf


def reason_layer(plan: CsrPlan, h, dist, ins, relfeat, relfeat_inv, W_rel, b_rel, W_e2e, b_e2e, w_score,
                 b_score, mask, pos=None, pos_inv=None, ws: Optional[LayerWorkspace] = None,
                 path: int = _lib.PATH_AUTO, math: Optional[int] = None):
    """One ReasonGNNLayer.forward (reasongnn.py:134-174) = ONE call into the library.
    Returns (h_out [B,N,D], score [B,N], dist_out [B,N])."""
    lib = _lib.load()
    B, N, R1 = plan.B, plan.N, plan.R1
    ins = _chk(ins, "ins")
    _, I, D = ins.shape
    h = _chk(h, "h").reshape(B * N, D)
    _on_plan_device(plan, h, "local_entity_emb")
    dist = _chk(dist, "dist").reshape(-1)
    mask = _chk(mask, "mask").reshape(-1)
    relfeat = _chk(relfeat, "rel_features", shape=(R1, D))
    relfeat_inv = _chk(relfeat_inv, "rel_features_inv", shape=(R1, D))
    W_rel = _chk(W_rel, "rel_linear.weight", shape=(D, D))
    b_rel = _chk(b_rel, "rel_linear.bias", shape=(D,))
    W_e2e = _chk(W_e2e, "e2e_linear.weight", shape=(D, (2 * I + 1) * D))
    b_e2e = _chk(b_e2e, "e2e_linear.bias", shape=(D,))
    w_score = _chk(w_score, "score_func.weight").reshape(-1)
    b_score = _chk(b_score, "score_func.bias").reshape(-1)
    if dist.numel() != B * N or mask.numel() != B * N or ins.shape[0] != B or w_score.numel() != D:
        raise ValueError("layer inputs do not match the plan (B=%d, N=%d, D=%d)" % (B, N, D))
    pos_rows = 0
    if pos is not None:
        pos = _chk(pos, "pos_emb.weight")
        pos_inv = _chk(pos_inv, "pos_emb_inv.weight", shape=tuple(pos.shape))
        pos_rows = pos.shape[0]
        if pos.shape[1] != D or pos_rows > R1:
            raise ValueError("pos_emb must be [<=R1, D]")
    ws = ws or LayerWorkspace()
    wbuf = ws.get(plan, D, I, h.device)
    h_out = torch.empty((B, N, D), dtype=torch.float32, device=h.device)
    score = torch.empty((B, N), dtype=torch.float32, device=h.device)
    dist_out = torch.empty((B, N), dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        _lib.check(lib.gnnrag_reason_layer(
            C.byref(plan.c), h.data_ptr(), dist.data_ptr(), ins.data_ptr(), relfeat.data_ptr(),
            relfeat_inv.data_ptr(), W_rel.data_ptr(), b_rel.data_ptr(), _ptr(pos), _ptr(pos_inv), pos_rows,
            W_e2e.data_ptr(), b_e2e.data_ptr(), w_score.data_ptr(), b_score.data_ptr(), mask.data_ptr(),
            h_out.data_ptr(), score.data_ptr(), dist_out.data_ptr(), wbuf.data_ptr(), wbuf.numel(), D, I,
            int(path), _math(math), _stream()), "gnnrag_reason_layer")
    return h_out, score, dist_out


class LayerStack:
    """The L ``ReasonGNNLayer.forward`` calls of one ReaRev iteration (rearev.py:208-210) as ONE library call
    (``gnnrag_reason_stack``), optionally captured as a hipGraph and replayed (``gnnrag_reason_stack_capture``).

    Everything that does not change between the iterations of a batch is validated and turned into raw pointers once,
    here: the structure, the relation features, the layers' parameters, the mask and the output buffers
    (``h [L,B,N,D]``, ``score`` / ``dist [L,B,N]``: every layer's outputs are kept, the reference returns each of
    them to its caller).  ``run(h0, dist0, ins)`` then costs one ctypes call."""

    def __init__(self, plan: CsrPlan, relfeat, relfeat_inv, layers, w_score, b_score, mask, I: int,
                 path: int = _lib.PATH_AUTO, math: Optional[int] = None):
        """layers: list of (W_rel, b_rel, W_e2e, b_e2e, pos, pos_inv) tensors per layer (pos / pos_inv may be None)."""
        lib = _lib.load()
        B, N, R1 = plan.B, plan.N, plan.R1
        relfeat = _chk(relfeat, "rel_features")
        D = relfeat.shape[1]
        _on_plan_device(plan, relfeat, "rel_features")
        self.plan, self.B, self.N, self.D, self.I, self.L = plan, B, N, D, int(I), len(layers)
        self.path, self.math = int(path), _math(math)
        keep = [relfeat, _chk(relfeat_inv, "rel_features_inv", shape=(R1, D)),
                _chk(w_score, "score_func.weight").reshape(-1), _chk(b_score, "score_func.bias").reshape(-1),
                _chk(mask, "mask").reshape(-1)]
        if tuple(relfeat.shape) != (R1, D) or keep[2].numel() != D or keep[4].numel() != B * N:
            raise ValueError("layer stack inputs do not match the plan (B=%d, N=%d, R1=%d, D=%d)" % (B, N, R1, D))
        self._relfeat, self._relfeat_inv, self._ws, self._bs, self._mask = keep
        self._params = (_lib.LayerParams * self.L)()
        self.pos_rows = 0
        for j, (W_rel, b_rel, W_e2e, b_e2e, pos, pos_inv) in enumerate(layers):
            t = [_chk(W_rel, "rel_linear.weight", shape=(D, D)), _chk(b_rel, "rel_linear.bias", shape=(D,)),
                 _chk(W_e2e, "e2e_linear.weight", shape=(D, (2 * self.I + 1) * D)),
                 _chk(b_e2e, "e2e_linear.bias", shape=(D,))]
            if pos is not None:
                pos = _chk(pos, "pos_emb.weight")
                pos_inv = _chk(pos_inv, "pos_emb_inv.weight", shape=tuple(pos.shape))
                if pos.shape[1] != D or pos.shape[0] > R1 or (self.pos_rows and pos.shape[0] != self.pos_rows):
                    raise ValueError("pos_emb must be [<=R1, D], the same size in every layer")
                self.pos_rows = pos.shape[0]
                t += [pos, pos_inv]
            keep += t
            pr = self._params[j]
            pr.W_rel, pr.b_rel, pr.W_e2e, pr.b_e2e = (x.data_ptr() for x in t[:4])
            pr.pos_fwd, pr.pos_inv = (t[4].data_ptr(), t[5].data_ptr()) if pos is not None else (None, None)
        self._keep = keep                                        # the tensors behind the raw pointers stay alive
        self.device = relfeat.device
        # the stack-sized workspace: relation projections of all L layers up front in one launch
        nbytes = max(lib.gnnrag_stack_workspace_bytes(C.byref(plan.c), self.L, D, self.I), 256)
        self._ws_buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._graph = self._graph_rest = None
        self.h = self.score = self.dist = None                   # graph mode: the fixed buffers of the captured sequence
        # the relation projections of the L layers depend on the parameters and the relation features only: the first
        # run of a forward computes them into the workspace, the runs of its later iterations reuse them
        # (GNNRAG_PATH_REUSE_PROJ).  A stack is bound to one batch (the module builds a new one per batch); a caller that
        # keeps one stack across forwards calls new_forward() at the start of each.
        self._proj_valid = False

    def new_forward(self):
        """The next run() recomputes the relation projections (start of a new forward / batch)."""
        self._proj_valid = False

    def _new_outputs(self):
        f32, dev, L, B, N, D = torch.float32, self.device, self.L, self.B, self.N, self.D
        return (torch.empty((L, B, N, D), dtype=f32, device=dev), torch.empty((L, B, N), dtype=f32, device=dev),
                torch.empty((L, B, N), dtype=f32, device=dev))

    def _args(self, h0, dist0, ins, out, reuse=False):
        h, score, dist = out
        return (C.byref(self.plan.c), self.L, self._params, h0.data_ptr(), dist0.data_ptr(), ins.data_ptr(),
                self._relfeat.data_ptr(), self._relfeat_inv.data_ptr(), self.pos_rows, self._ws.data_ptr(),
                self._bs.data_ptr(), self._mask.data_ptr(), h.data_ptr(), score.data_ptr(),
                dist.data_ptr(), self._ws_buf.data_ptr(), self._ws_buf.numel(), self.D, self.I,
                self.path | (_lib.PATH_REUSE_PROJ if reuse else 0), self.math)

    def _inputs(self, h0, dist0, ins):
        h0 = _chk(h0, "local_entity_emb").reshape(self.B * self.N, self.D)
        dist0 = _chk(dist0, "dist").reshape(-1)
        ins = _chk(ins, "relational_ins", shape=(self.B, self.I, self.D))
        if dist0.numel() != self.B * self.N:
            raise ValueError("dist does not match the plan")
        _on_plan_device(self.plan, h0, "local_entity_emb")
        return h0, dist0, ins

    def run(self, h0, dist0, ins):
        """Runs the L layers; returns freshly allocated (h [L,B,N,D], score [L,B,N], dist [L,B,N])."""
        h0, dist0, ins = self._inputs(h0, dist0, ins)
        out = self._new_outputs()
        reuse = self._proj_valid and self.L > 1
        with torch.cuda.device(h0.device):
            _lib.check(_lib.load().gnnrag_reason_stack(*self._args(h0, dist0, ins, out, reuse), _stream()),
                       "gnnrag_reason_stack")
        self._proj_valid = True
        return out

    def capture(self, h0, dist0, ins):
        """Captures the sequence as a hipGraph over FIXED buffers (``self.h / score / dist``, allocated here): the
        node state is read from ``self.h[L-1]`` (the previous iteration's last layer; ``h0`` is copied there now),
        the prior from ``dist0`` and the instructions from ``ins`` - both are kept and read again by every replay, so
        rewrite them in place between replays (rearev.py:208,217-221).  An eager ``run`` must have happened before
        (launch attributes are raised on first use)."""
        if self.L < 2:
            raise ValueError("graph replay needs num_gnn >= 2 (layer 0 reads the buffer the last layer writes)")
        self.release_graph()
        self.h, self.score, self.dist = self._new_outputs()
        self.h[self.L - 1].copy_(h0.reshape(self.B, self.N, self.D))
        h0g, dist0, ins = self._inputs(self.h[self.L - 1], dist0, ins)
        g, g2 = C.c_void_p(), C.c_void_p()
        with torch.cuda.device(h0g.device):
            # stream capture is not allowed on the legacy default stream (torch's default): capture on a side stream
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.check(_lib.load().gnnrag_reason_stack_capture(
                    *self._args(h0g, dist0, ins, (self.h, self.score, self.dist)), _stream(), C.byref(g)),
                    "gnnrag_reason_stack_capture")
                # second graph: the same sequence WITHOUT the relation projections - the iterations 2..T of a forward
                # (the eager run every capture needs left them in the workspace; every replay of the first graph
                # rewrites them)
                _lib.check(_lib.load().gnnrag_reason_stack_capture(
                    *self._args(h0g, dist0, ins, (self.h, self.score, self.dist), reuse=True), _stream(), C.byref(g2)),
                    "gnnrag_reason_stack_capture")
            torch.cuda.current_stream().wait_stream(side)
        self._graph, self._graph_rest, self._graph_in = g, g2, (dist0, ins)

    def capture_rest(self, h_prev, dist0, ins):
        """Module path (``ReasonGNNLayer._forward_stack``, iterations 2..T of a forward): captures ONLY the sequence without
        the relation projections (the eager run of iteration 1 left them in the workspace) over fixed buffers - the node
        state in ``self.h[L-1]`` (``h_prev`` is copied there), the prior ``dist0`` (the reference hands the same seed
        tensor to every iteration, rearev.py:208) and an instruction buffer of this stack, refreshed by ``replay_rest``."""
        if self.L < 2:
            raise ValueError("graph replay needs num_gnn >= 2 (layer 0 reads the buffer the last layer writes)")
        if not self._proj_valid:
            raise RuntimeError("capture_rest() needs the projections of an eager run() in the workspace")
        self.release_graph()
        self.h, self.score, self.dist = self._new_outputs()
        self._ins_buf = torch.empty((self.B, self.I, self.D), dtype=torch.float32, device=self.device)
        self.h[self.L - 1].copy_(h_prev.reshape(self.B, self.N, self.D))
        self._ins_buf.copy_(ins)
        h0g, dist0, insb = self._inputs(self.h[self.L - 1], dist0, self._ins_buf)
        g2 = C.c_void_p()
        with torch.cuda.device(h0g.device):
            side = torch.cuda.Stream()                            # no capture on the legacy default stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.check(_lib.load().gnnrag_reason_stack_capture(
                    *self._args(h0g, dist0, insb, (self.h, self.score, self.dist), reuse=True), _stream(), C.byref(g2)),
                    "gnnrag_reason_stack_capture")
            torch.cuda.current_stream().wait_stream(side)
        self._graph_rest, self._graph_in = g2, (dist0, insb)

    def replay_rest(self, h_prev, dist0, ins):
        """One replay of the graph of ``capture_rest``: ``h_prev`` must be the previous replay's / capture's last-layer
        state (``self.h[L-1]``; anything else is copied in), ``dist0`` the captured prior tensor, ``ins`` is copied into the
        captured instruction buffer.  Returns the fixed buffers (overwritten by the next replay)."""
        if getattr(self, "_graph_rest", None) is None:
            raise RuntimeError("capture_rest() first")
        if dist0.data_ptr() != self._graph_in[0].data_ptr():
            raise RuntimeError("replay_rest: the prior is not the captured tensor")
        if h_prev.data_ptr() != self.h[self.L - 1].data_ptr():
            self.h[self.L - 1].copy_(h_prev.reshape(self.B, self.N, self.D))
        self._ins_buf.copy_(ins)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gnnrag_graph_launch(self._graph_rest, _stream()), "gnnrag_graph_launch")
        return self.h, self.score, self.dist

    def replay(self, first: bool = True):
        """Replays the captured sequence.  ``first=False``: the graph without the relation-projection launch (the later
        iterations of a forward; the first one's replay left the projections in the workspace)."""
        if self._graph is None:
            raise RuntimeError("capture() first")
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gnnrag_graph_launch(self._graph if first or self.L < 2 else self._graph_rest, _stream()),
                       "gnnrag_graph_launch")
        return self.h, self.score, self.dist

    def release_graph(self):
        for name in ("_graph", "_graph_rest"):
            if getattr(self, name, None) is not None:
                _lib.load().gnnrag_graph_destroy(getattr(self, name))
                setattr(self, name, None)

    def __del__(self):
        try:
            self.release_graph()
        except Exception:
            pass


def lstm_forward(x, w_ih, w_hh, b_ih=None, b_hh=None, h0=None, c0=None, workspaces=None):
    """One-layer batch_first LSTM, torch.nn.LSTM semantics (lstm_encoder.py:27-36): x [B,T,E], w_ih [4H,E], w_hh [4H,H],
    biases [4H] or None, h0 / c0 [B,H] or None (zeros).  Returns (out [B,T,H], h_n [B,H], c_n [B,H]).
    ``workspaces``: a dict OWNED BY THE CALLER (``HipLSTM`` keeps one per module) in which the transposed-weight scratch
    is kept between calls, keyed by (device, stream); None = a fresh scratch for this call."""
    lib = _lib.load()
    x = _chk(x, "x")
    if x.dim() != 3:
        raise ValueError("x must be [B,T,E]")
    B, T, E = x.shape
    w_ih = _chk(w_ih.detach(), "w_ih")
    H = w_ih.shape[0] // 4
    w_ih = _chk(w_ih, "w_ih", shape=(4 * H, E))
    w_hh = _chk(w_hh.detach(), "w_hh", shape=(4 * H, H))
    b_ih = None if b_ih is None else _chk(b_ih.detach(), "b_ih", shape=(4 * H,))
    b_hh = None if b_hh is None else _chk(b_hh.detach(), "b_hh", shape=(4 * H,))
    h0 = None if h0 is None else _chk(h0, "h0", shape=(B, H))
    c0 = None if c0 is None else _chk(c0, "c0", shape=(B, H))
    dev = x.device
    out = torch.empty((B, T, H), dtype=torch.float32, device=dev)
    h_n = torch.empty((B, H), dtype=torch.float32, device=dev)
    c_n = torch.empty((B, H), dtype=torch.float32, device=dev)
    need = lib.gnnrag_lstm_workspace_bytes(E, H)
    # one workspace per (device, stream): two calls on different streams of one device must not share the transposed-weight
    # scratch (calls on one stream are ordered)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = workspaces.get(key) if workspaces is not None else None
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        if workspaces is not None:
            workspaces[key] = ws
    with torch.cuda.device(dev):
        _lib.check(lib.gnnrag_lstm_forward(x.data_ptr(), w_ih.data_ptr(), w_hh.data_ptr(), _ptr(b_ih), _ptr(b_hh), _ptr(h0),
                                           _ptr(c0), out.data_ptr(), h_n.data_ptr(), c_n.data_ptr(), B, T, E, H,
                                           ws.data_ptr(), ws.numel(), _stream()), "gnnrag_lstm_forward")
    return out, h_n, c_n


def seed_retrieve(seed_info: torch.Tensor, ent_emb: torch.Tensor) -> torch.Tensor:
    """sum_n seed_info[b,n] * ent_emb[b,n,:]  ->  [B,D] (query_update.py:40), reading only flagged rows."""
    lib = _lib.load()
    seed_info = _chk(seed_info, "seed_info")
    B, N = seed_info.shape
    ent_emb = _chk(ent_emb, "ent_emb")
    if ent_emb.dim() != 3 or ent_emb.shape[0] != B or ent_emb.shape[1] != N:
        raise ValueError("ent_emb must be [B,N,D] matching seed_info [B,N]")
    D = ent_emb.shape[2]
    out = torch.empty((B, D), dtype=torch.float32, device=ent_emb.device)
    with torch.cuda.device(ent_emb.device):
        _lib.check(lib.gnnrag_seed_retrieve(seed_info.data_ptr(), ent_emb.data_ptr(), out.data_ptr(), B, N, D,
                                            _stream()), "gnnrag_seed_retrieve")
    return out


def query_reform(q_node: torch.Tensor, seed_info: torch.Tensor, ent_emb: torch.Tensor, W_r: torch.Tensor,
                 W_g: torch.Tensor) -> torch.Tensor:
    """``QueryReform.forward`` in one launch (query_update.py:26-44 with Fusion :6-16):
    ``fusion(q_node, seed_retrieve(seed_info, ent_emb))`` -> [B, D].  ``ent_emb`` [B, N, D'] with D' >= D = q_node's
    width: a zero-padded node state is read in place."""
    lib = _lib.load()
    q_node = _chk(q_node, "q_node")
    seed_info = _chk(seed_info, "seed_info")
    ent_emb = _chk(ent_emb, "ent_emb")
    W_r = _chk(W_r, "W_r")
    W_g = _chk(W_g, "W_g")
    B, D = q_node.shape
    N = seed_info.shape[1]
    if (seed_info.shape[0] != B or ent_emb.dim() != 3 or ent_emb.shape[0] != B or ent_emb.shape[1] != N
            or ent_emb.shape[2] < D):
        raise ValueError("query_reform: q_node [B,D], seed_info [B,N], ent_emb [B,N,>=D]")
    if tuple(W_r.shape) != (D, 3 * D) or tuple(W_g.shape) != (D, 3 * D):
        raise ValueError("query_reform: fusion weights must be [D, 3D]")
    out = torch.empty((B, D), dtype=torch.float32, device=q_node.device)
    with torch.cuda.device(q_node.device):
        _lib.check(lib.gnnrag_query_reform(q_node.data_ptr(), seed_info.data_ptr(), ent_emb.data_ptr(), ent_emb.shape[2],
                                           W_r.data_ptr(), W_g.data_ptr(), out.data_ptr(), B, N, D, _stream()),
                   "gnnrag_query_reform")
    return out


def topp_candidates(pred_dist: torch.Tensor, eligible: torch.Tensor, ignore_prob: float, eps: float):
    """Per question: slots kept by the Evaluator's filter, sorted by probability (descending, stable), and
    how many of them the top-p cut retrieves.  Returns (slots int32 [B,N] (-1 padded), counts int32 [B,2])."""
    lib = _lib.load()
    pred_dist = _chk(pred_dist, "pred_dist")
    B, N = pred_dist.shape
    eligible = _chk(eligible, "eligible", dtype=torch.uint8, shape=(B, N))
    slots = torch.empty((B, N), dtype=torch.int32, device=pred_dist.device)
    cnt = torch.empty((B, 2), dtype=torch.int32, device=pred_dist.device)
    with torch.cuda.device(pred_dist.device):
        nws = lib.gnnrag_topp_workspace_bytes(B, N)          # > 0 for N > 16384: the survivors are sorted outside LDS
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=pred_dist.device)
        _lib.check(lib.gnnrag_topp_candidates_ws(pred_dist.data_ptr(), eligible.data_ptr(), B, N, float(ignore_prob),
                                                 float(eps), slots.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _stream()), "gnnrag_topp_candidates_ws")
    return slots, cnt


def stream_copy(src: torch.Tensor, dst: torch.Tensor):
    lib = _lib.load()
    with torch.cuda.device(src.device):
        _lib.check(lib.gnnrag_stream_copy(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()),
                   "gnnrag_stream_copy")
