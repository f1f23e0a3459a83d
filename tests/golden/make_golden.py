#!/usr/bin/env python
"""Generates tests/golden/*.npz from the LIVE reference (cmavro/GNN-RAG @ v2).

Run in the build container only (needs /root/reference, CPU is enough):

    python tests/golden/make_golden.py

The reference ships no tests or golden vectors for the reasoning hot path
(SURVEY.md section 4), so the fixtures are made by importing the reference's own
modules - ``modules.kg_reasoning.reasongnn.ReasonGNNLayer``,
``modules.layer_init.TypeLayer`` and, for the call-site anchored case,
the full ``models.ReaRev.rearev.ReaRev`` driven through ``dataset_load`` - and
recording their inputs and outputs.  Nothing from the reference is copied into
the repository: only numeric inputs/outputs are stored.

Fixtures (all float32 / int64, np.savez_compressed):
  layer_d200.npz   ReasonGNNLayer, D=200 I=2 L=3 T=2, unnormalised, no pos_emb
  layer_d50.npz    ReasonGNNLayer, D=50 I=3 L=2 T=2, normalized_gnn + pos_emb,
                   ragged questions incl. a 1-node question and a hub node
  typelayer.npz    TypeLayer on the layer_d50 graph, norm_rel False and True
  rearev_e2e.npz   layer-boundary tensors captured inside a real
                   ReaRev.forward on a synthetic on-disk dataset (LSTM encoder)
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/gnn"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import gnnrag_amd  # noqa: E402
from gnnrag_amd import synth  # noqa: E402


def layer_args(cfg):
    return dict(use_cuda=False, normalized_gnn=cfg.normalized_gnn, num_ins=cfg.I,
                num_gnn=cfg.L, pos_emb=cfg.pos_emb, linear_dropout=0.0)


def edge_arrays(edge_tuple):
    h, r, t, b, f, wl, wrl = edge_tuple
    return dict(heads=np.asarray(h, np.int64), rels=np.asarray(r, np.int64),
                tails=np.asarray(t, np.int64), batch_ids=np.asarray(b, np.int64),
                weight_list=np.asarray(wl, np.float64),
                weight_rel_list=np.asarray(wrl, np.float64))


def run_reference_layer(cfg, batch, feats, params):
    from modules.kg_reasoning.reasongnn import ReasonGNNLayer
    layer = ReasonGNNLayer(layer_args(cfg), batch.num_entity, cfg.num_kb_relation, cfg.D, "bfs")
    sd = {k: torch.from_numpy(v) for k, v in params.items() if not k.startswith("type_layer.")}
    layer.load_state_dict(sd, strict=True)
    layer.eval()
    out = {"score": [], "dist": [], "h": []}
    with torch.no_grad():
        local_entity = torch.from_numpy(batch.local_entity)
        layer.init_reason(local_entity=local_entity, kb_adj_mat=batch.edge_tuple,
                          local_entity_emb=torch.from_numpy(feats["h0"]),
                          rel_features=torch.from_numpy(feats["rel_features"]),
                          rel_features_inv=torch.from_numpy(feats["rel_features_inv"]),
                          query_entities=torch.from_numpy(batch.query_entities).float())
        seed = torch.from_numpy(batch.seed_dist).float()
        for t in range(cfg.T):
            dist = seed
            ins = torch.from_numpy(feats["ins"][t])
            for j in range(cfg.L):
                score, dist = layer(dist, ins, step=j, return_score=True)
                out["score"].append(score.numpy().copy())
                out["dist"].append(dist.numpy().copy())
                out["h"].append(layer.local_entity_emb.numpy().copy())
    return out


def save_layer_case(name, cfg, batch, feats, params, ref_out):
    d = dict(B=cfg.B, N=cfg.N, D=cfg.D, I=cfg.I, L=cfg.L, T=cfg.T, R1=cfg.R1,
             num_kb_relation=cfg.num_kb_relation, num_entity=batch.num_entity,
             normalized_gnn=int(cfg.normalized_gnn), pos_emb=int(cfg.pos_emb),
             local_entity=batch.local_entity, query_entities=batch.query_entities,
             seed_dist=batch.seed_dist, n_real=batch.n_real)
    d.update(edge_arrays(batch.edge_tuple))
    d.update({"feat." + k: v for k, v in feats.items()})
    d.update({"param." + k: v for k, v in params.items()})
    for k, lst in ref_out.items():
        d["ref." + k] = np.stack(lst)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print("wrote", name, {k: d["ref." + k].shape for k in ref_out})


def case_d200():
    cfg = synth.GraphConfig(name="gold200", B=3, N=40, E=110, R=9, D=200, I=2, L=3, T=2, seed=11)
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    save_layer_case("layer_d200.npz", cfg, batch, feats, params,
                    run_reference_layer(cfg, batch, feats, params))


def ragged_batch(cfg, n_real, seed):
    """Like synth.make_batch but with hand-picked real-node counts (edge cases)."""
    rng = np.random.default_rng(seed)
    n_real = np.asarray(n_real, np.int64)
    num_entity = 1000
    local_entity = np.full((cfg.B, cfg.N), num_entity, np.int64)
    for i in range(cfg.B):
        local_entity[i, : n_real[i]] = rng.integers(0, num_entity, size=n_real[i])
    q = np.zeros((cfg.B, cfg.N))
    q[:, 0] = 1.0
    local_entity[:, 0] = num_entity
    # question 3 keeps its seed as a valid answer slot (CWQ-style, dataset_load.py:249-257)
    local_entity[3, 0] = 7
    et = synth.make_edge_tuple(cfg, rng, n_real)
    return synth.Batch(cfg=cfg, local_entity=local_entity, query_entities=q, seed_dist=q.copy(),
                       edge_tuple=et, num_entity=num_entity, n_real=n_real)


def case_d50_and_typelayer():
    cfg = synth.GraphConfig(name="gold50", B=4, N=36, E=80, R=7, D=50, I=3, L=2, T=2,
                            normalized_gnn=True, pos_emb=True, seed=12)
    batch = ragged_batch(cfg, [36, 5, 1, 20], seed=12)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    save_layer_case("layer_d50.npz", cfg, batch, feats, params,
                    run_reference_layer(cfg, batch, feats, params))

    from modules.layer_init import TypeLayer
    res = {}
    for norm_rel in (False, True):
        tl = TypeLayer(cfg.D, cfg.D, torch.nn.Dropout(0.0), torch.device("cpu"), norm_rel)
        tl.load_state_dict({"kb_self_linear.weight": torch.from_numpy(params["type_layer.kb_self_linear.weight"]),
                            "kb_self_linear.bias": torch.from_numpy(params["type_layer.kb_self_linear.bias"])})
        with torch.no_grad():
            h0 = tl(local_entity=torch.from_numpy(batch.local_entity), edge_list=batch.edge_tuple,
                    rel_features=torch.from_numpy(feats["rel_features"]))
        res["ref.h0_norm%d" % int(norm_rel)] = h0.numpy().copy()
    d = dict(B=cfg.B, N=cfg.N, D=cfg.D, R1=cfg.R1, num_entity=batch.num_entity,
             local_entity=batch.local_entity)
    d.update(edge_arrays(batch.edge_tuple))
    d["feat.rel_features"] = feats["rel_features"]
    d["param.type_layer.kb_self_linear.weight"] = params["type_layer.kb_self_linear.weight"]
    d["param.type_layer.kb_self_linear.bias"] = params["type_layer.kb_self_linear.bias"]
    d.update(res)
    np.savez_compressed(os.path.join(HERE, "typelayer.npz"), **d)
    print("wrote typelayer.npz")


# ----------------------------------------------------------------------------------------
# call-site anchored case: a real ReaRev.forward on an on-disk synthetic dataset
# ----------------------------------------------------------------------------------------
def write_dataset(folder, rng, n_ent=120, n_rel=9, n_q=6):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "entities.txt"), "w") as f:
        for i in range(n_ent):
            f.write("m.%04d\n" % i)
    with open(os.path.join(folder, "relations.txt"), "w") as f:
        for i in range(n_rel):
            f.write("rel.type%d.name%d\n" % (i % 3, i))
    words = ["what", "is", "the", "name", "of", "who", "where", "film", "city", "born"]
    with open(os.path.join(folder, "vocab.txt"), "w") as f:
        for w in words:
            f.write(w + "\n")
    for split in ("train", "dev", "test"):
        with open(os.path.join(folder, split + ".json"), "w") as f:
            for qi in range(n_q):
                n_sub = int(rng.integers(8, 30))
                ents = rng.choice(n_ent, size=n_sub, replace=False).tolist()
                tuples = []
                for _ in range(int(rng.integers(n_sub, 3 * n_sub))):
                    h, t = rng.choice(ents, size=2)
                    tuples.append([int(h), int(rng.integers(0, n_rel)), int(t)])
                seed_e = ents[0]
                ans = ents[1 + int(rng.integers(0, n_sub - 1))]
                q = " ".join(rng.choice(words, size=int(rng.integers(3, 7))).tolist())
                f.write(json.dumps({
                    "id": "%s-%d" % (split, qi), "question": q, "entities": [seed_e],
                    "answers": [{"kb_id": "m.%04d" % ans, "text": "a"}],
                    "subgraph": {"tuples": tuples, "entities": ents}}) + "\n")


def case_rearev_e2e():
    import parsing
    parsing.create_parser_nutrea = lambda p: None            # reference bug, SURVEY section 4(1)
    from modules.question_encoding import base_encoder
    _orig_init = base_encoder.BaseInstruction.__init__

    def _init(self, args, constraint=False):                  # reference bug, SURVEY section 4(2)
        _orig_init(self, args, constraint)
    base_encoder.BaseInstruction.__init__ = _init

    rng = np.random.default_rng(5)
    tmp = tempfile.mkdtemp(prefix="gnnrag_gold_")
    folder = os.path.join(tmp, "synth") + "/"
    write_dataset(folder, rng)
    D = 50
    argv = ["ReaRev", "--data_folder", folder, "--lm", "lstm", "--relation_word_emb", "False",
            "--entity_dim", str(D), "--kg_dim", str(D // 2), "--word_dim", "24",
            "--num_iter", "2", "--num_ins", "2", "--num_gnn", "3", "--batch_size", "3",
            "--test_batch_size", "3", "--checkpoint_dir", tmp + "/", "--experiment_name", "gold",
            "--name", "synth"]
    import argparse
    parser = argparse.ArgumentParser()
    parsing.add_parse_args(parser)
    args = vars(parser.parse_args(argv))
    args["use_cuda"] = False
    args["word_emb_file"] = None          # no pretrained word vectors in the container
    np.random.seed(args["seed"])
    torch.manual_seed(args["seed"])

    from dataset_load import load_data
    from models.ReaRev.rearev import ReaRev
    dataset = load_data(args, args["lm"])
    test = dataset["test"]
    model = ReaRev(args, len(dataset["entity2id"]), test.num_kb_relation, dataset["num_word"])
    model.eval()

    rec = {"calls": []}
    reasoning = model.reasoning
    orig_init_reason = reasoning.init_reason
    orig_forward = reasoning.forward

    def init_reason_hook(**kw):
        rec["init"] = {k: (v.detach().numpy().copy() if torch.is_tensor(v) else v) for k, v in kw.items()}
        return orig_init_reason(**kw)

    def forward_hook(current_dist, relational_ins, step=0, return_score=False):
        res = orig_forward(current_dist, relational_ins, step=step, return_score=return_score)
        rec["calls"].append(dict(step=step, dist_in=current_dist.detach().numpy().copy(),
                                 ins=relational_ins.detach().numpy().copy(),
                                 dist_out=res[0].detach().numpy().copy(),
                                 h_out=res[1].detach().numpy().copy()))
        return res

    reasoning.init_reason = init_reason_hook
    reasoning.forward = forward_hook
    test.reset_batches(is_sequential=True)
    np.random.seed(123)
    batch = test.get_batch(0, 3, fact_dropout=0.0, test=True)
    with torch.no_grad():
        loss, pred, pred_dist, _ = model(batch[:-1])

    init = rec["init"]
    d = dict(B=batch[0].shape[0], N=batch[0].shape[1], D=D, I=2, L=3, T=2,
             num_entity=len(dataset["entity2id"]), num_kb_relation=test.num_kb_relation,
             local_entity=init["local_entity"], h0=init["local_entity_emb"],
             rel_features=init["rel_features"], rel_features_inv=init["rel_features_inv"],
             query_entities=init["query_entities"], pred=pred.numpy(), pred_dist=pred_dist.numpy())
    d.update(edge_arrays(init["kb_adj_mat"]))
    for k, v in reasoning.state_dict().items():
        d["param." + k] = v.numpy().copy()
    d["call.step"] = np.array([c["step"] for c in rec["calls"]])
    for key in ("dist_in", "ins", "dist_out", "h_out"):
        d["call." + key] = np.stack([c[key] for c in rec["calls"]])
    np.savez_compressed(os.path.join(HERE, "rearev_e2e.npz"), **d)
    print("wrote rearev_e2e.npz: %d layer calls, B=%d N=%d" % (len(rec["calls"]), d["B"], d["N"]))


if __name__ == "__main__":
    torch.manual_seed(0)
    case_d200()
    case_d50_and_typelayer()
    case_rearev_e2e()
