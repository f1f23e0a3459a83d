"""Host-side driver of the layer stack, the way ``ReaRev.forward`` drives it
(reference ``gnn/models/ReaRev/rearev.py:206-211``): for each of T iterations reset
``dist`` to the seed distribution, run L ``ReasonGNNLayer.forward`` calls, node
embeddings carry over.  Used by the parity tests, ``smoke()`` and ``bench.py`` with
synthetic inputs (``synth.Batch`` / ``make_features`` / ``make_layer_params``)."""
from __future__ import annotations

import numpy as np
import torch

from .modules.kg_reasoning.reasongnn import ReasonGNNLayer
from .modules.layer_init import TypeLayer


def layer_args(cfg, use_cuda=True) -> dict:
    return dict(use_cuda=use_cuda, normalized_gnn=cfg.normalized_gnn, num_ins=cfg.I, num_gnn=cfg.L,
                pos_emb=cfg.pos_emb, linear_dropout=0.0)


def build_layer(cfg, batch, params: dict, device, path: int = None) -> ReasonGNNLayer:
    layer = ReasonGNNLayer(layer_args(cfg), batch.num_entity, cfg.num_kb_relation, cfg.D, "bfs")
    if path is not None:
        layer.path = path
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in params.items() if not k.startswith("type_layer.")}
    layer.load_state_dict(sd, strict=True)
    return layer.to(device).eval()


def build_type_layer(cfg, params: dict, device, norm_rel: bool) -> TypeLayer:
    tl = TypeLayer(cfg.D, cfg.D, torch.nn.Dropout(0.0), device, norm_rel)
    tl.load_state_dict({"kb_self_linear.weight": torch.from_numpy(params["type_layer.kb_self_linear.weight"]),
                        "kb_self_linear.bias": torch.from_numpy(params["type_layer.kb_self_linear.bias"])})
    return tl.to(device).eval()


class DeviceInputs:
    """The dense inputs of one batch, resident in HBM."""

    def __init__(self, batch, feats: dict, device):
        f32 = torch.float32
        self.local_entity = torch.from_numpy(batch.local_entity).to(device)
        self.query_entities = torch.from_numpy(batch.query_entities).to(device, f32)
        self.seed_dist = torch.from_numpy(batch.seed_dist).to(device, f32)      # rearev.py:174
        self.rel_features = torch.from_numpy(feats["rel_features"]).to(device)
        self.rel_features_inv = torch.from_numpy(feats["rel_features_inv"]).to(device)
        self.ins = torch.from_numpy(feats["ins"]).to(device)                     # [T,B,I,D]
        self.h0 = torch.from_numpy(feats["h0"]).to(device) if "h0" in feats else None


@torch.no_grad()
def init_reason(layer: ReasonGNNLayer, batch, dev: DeviceInputs, h0: torch.Tensor):
    layer.init_reason(local_entity=dev.local_entity, kb_adj_mat=batch.edge_tuple, local_entity_emb=h0,
                      rel_features=dev.rel_features, rel_features_inv=dev.rel_features_inv,
                      query_entities=dev.query_entities)


@torch.no_grad()
def run_layers(layer: ReasonGNNLayer, cfg, dev: DeviceInputs, record: bool = False):
    """T x L layer calls on an initialised layer.  Returns the last dist (and a record)."""
    rec = {"score": [], "dist": [], "h": []} if record else None
    dist = dev.seed_dist
    for t in range(cfg.T):
        dist = dev.seed_dist                                    # rearev.py:208
        ins = dev.ins[t]
        for j in range(cfg.L):                                  # rearev.py:209-210
            if record:
                score, dist = layer(dist, ins, step=j, return_score=True)
                rec["score"].append(score.cpu().numpy())
                rec["dist"].append(dist.cpu().numpy())
                rec["h"].append(layer.local_entity_emb.cpu().numpy())
            else:
                dist, _ = layer(dist, ins, step=j)
    return dist, rec


@torch.no_grad()
def run_stack(batch, feats: dict, params: dict, device, *, use_type_layer=False, norm_rel=False,
              path: int = None):
    """Mirror of ``oracle.*.run_stack`` on the HIP path."""
    cfg = batch.cfg
    dev = DeviceInputs(batch, feats, device)
    layer = build_layer(cfg, batch, params, device, path)
    out = {}
    if use_type_layer:
        tl = build_type_layer(cfg, params, device, norm_rel)
        h0 = tl(local_entity=dev.local_entity, edge_list=batch.edge_tuple, rel_features=dev.rel_features)
        out["h0"] = h0.cpu().numpy()
    else:
        h0 = dev.h0
    init_reason(layer, batch, dev, h0)
    _, rec = run_layers(layer, cfg, dev, record=True)
    out.update(rec)
    return out


@torch.no_grad()
def run_rearev_loop(type_layer, reasoning, reforms, *, local_entity, query_entities, edge_tuple, seed_dist,
                    rel_features, rel_features_inv, instructions, num_iter: int):
    """The reasoning part of ``ReaRev.forward`` (rearev.py:163-243) on built modules, with the tensors the encoders
    hand over (relation features, the initial instructions [B, I, D]) as inputs: TypeLayer start (``get_ent_init``,
    :79-88), ``init_reason`` (:147-153), ``num_iter`` x (``num_gnn`` layer calls, then one QueryReform per
    instruction: :206-221).  Nothing is fed back from outside between the calls.  Returns (pred, pred_dist)."""
    h0 = type_layer(local_entity=local_entity, edge_list=edge_tuple, rel_features=rel_features)
    reasoning.init_reason(local_entity=local_entity, kb_adj_mat=edge_tuple, local_entity_emb=h0,
                          rel_features=rel_features, rel_features_inv=rel_features_inv, query_entities=query_entities)
    ins = [instructions[:, j].unsqueeze(1) for j in range(instructions.shape[1])]      # rearev.py:194-196
    dist = seed_dist
    for _ in range(num_iter):
        relation_ins = torch.cat(ins, dim=1)                                          # :207
        dist = seed_dist                                                              # :208
        for j in range(reasoning.num_gnn):
            dist, global_rep = reasoning(dist, relation_ins, step=j)                  # :209-210
        for j, reform in enumerate(reforms):                                          # :217-221
            q = reform(ins[j].squeeze(1), global_rep, query_entities, local_entity)
            ins[j] = q.unsqueeze(1)
    return torch.max(dist, dim=1)[1], dist                                            # :236-237
