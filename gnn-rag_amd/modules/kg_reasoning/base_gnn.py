"""Drop-in for the reference's ``modules/kg_reasoning/base_gnn.py``.

``BaseGNNLayer.build_matrix`` (reference ``base_gnn.py:19-51``) turns the batch tuple
into seven COO tensors through Python lists.  Here it uploads the three int arrays once
and builds a destination-sorted structure on the GPU (``ops.CsrPlan`` ->
``gnnrag_csr_build``).  Same class name, constructor and attributes the callers read.
"""
from __future__ import annotations

import os

import torch

from ... import ops
from ..._lib import GnnragError

VERY_NEG_NUMBER = -100000000000

# ReaRev hands the SAME kb_adj_mat tuple object to TypeLayer and to the reasoning layer
# (rearev.py:81-83,147-153); keep the last plan so the batch is sorted once, not twice.
_last_plan = {"key": None, "plan": None, "tuple": None}


def plan_for(edge_list, B: int, N: int, R1: int, device) -> "ops.CsrPlan":
    key = (id(edge_list), B, N, R1, str(device))
    if _last_plan["key"] == key and _last_plan["tuple"] is edge_list:
        return _last_plan["plan"]
    plans = getattr(edge_list, "plans", None)             # data/fact_mat.DeviceStructureCache: per-question structures
    hrt = None if plans is not None else getattr(edge_list, "hrt_device", None)   # BatchFacts: the id block is on the GPU
    if plans is not None and len(plans) == B and all(p.N == N and p.R1 == R1 for p in plans):
        plan = ops.CsrPlan.concat(plans, N, R1, plans[0].device)          # copies with offsets: no sort, no sync
    elif plans is not None:
        hrt = edge_list.hrt_device
        plan = ops.CsrPlan(None, None, None, B, N, R1, hrt.device, hrt_device=hrt)
    elif hrt is not None:
        # a device fact cache knows every question's relation count: the build is told them and does not wait
        rc = getattr(edge_list, "rel_counts", None) if R1 > 0 else None
        # ... but only for the shape the cache's host-side range check covered (a caller with another N, or fewer table
        # rows than the largest relation id admits, gets the waiting build and its device-side validation instead of
        # undersized relation tables)
        cn, cr = getattr(edge_list, "checked_n", None), getattr(edge_list, "checked_r1", None)
        if rc is not None and (cn != N or cr is None or R1 < cr):
            rc = None
        plan = ops.CsrPlan(None, None, None, B, N, R1, hrt.device, hrt_device=hrt, rel_counts=rc)
        owner = getattr(edge_list, "owner", None)
        if rc is not None and (owner is None or not getattr(owner, "_status_checked", False)
                               or os.environ.get("GNNRAG_CHECK_STRUCTURES") == "1"):
            # the deferred device-side validation, ONCE per cache (= per loader split), or on every batch under
            # GNNRAG_CHECK_STRUCTURES=1: the counts told to the build are compared with what the device counted
            plan.status()
            if owner is not None:
                owner._status_checked = True
    else:
        plan = ops.CsrPlan(edge_list[0], edge_list[1], edge_list[2], B, N, R1, device)
    _last_plan.update(key=key, plan=plan, tuple=edge_list)
    return plan


def _device_from_args(args, like: torch.Tensor = None) -> torch.device:
    if not args.get("use_cuda", False):
        raise GnnragError("gnnrag_amd runs on the GPU only (args['use_cuda'] is False); "
                          "use the reference modules for a CPU run")
    if not torch.cuda.is_available():
        raise GnnragError("no ROCm device visible to torch; gnnrag_amd has no CPU fallback")
    if like is not None and like.is_cuda:
        return like.device          # the device the model's tensors live on, not whatever is current
    return torch.device("cuda", torch.cuda.current_device())


def _check_gpu_tensor(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise GnnragError("%s needs GPU tensors (got %s); gnnrag_amd has no CPU fallback" % (what, t.device))


class BaseGNNLayer(torch.nn.Module):
    """Builds the sparse structure of a batch (reference: base_gnn.py:9-54)."""

    def __init__(self, args, num_entity, num_relation):
        super().__init__()
        self.num_relation = num_relation
        self.num_entity = num_entity
        self._args_use_cuda = bool(args.get("use_cuda", False))
        self.device = torch.device("cuda" if self._args_use_cuda else "cpu")
        self.normalized_gnn = args["normalized_gnn"]
        self._args = dict(use_cuda=self._args_use_cuda)

    def build_matrix(self):
        """Same inputs as the reference method (``self.edge_list``, ``self.batch_size``,
        ``self.max_local_entity``, ``self.num_relation`` = rows of the relation tables,
        set by ``init_reason``); result is ``self.plan``."""
        device = _device_from_args(self._args, getattr(self, "rel_features", None))
        edge_list = self.edge_list
        self.num_fact = len(edge_list[4])
        self.plan = plan_for(edge_list, self.batch_size, self.max_local_entity, self.num_relation, device)
        if self.normalized_gnn:
            self.plan.attach_w_gnn(edge_list[5])
