#!/usr/bin/env python
"""Closed-loop fixture from the LIVE reference (build container only): a multi-batch ``Evaluator.evaluate`` run of
the reference's own ReaRev (gnn/models/ReaRev/rearev.py:163-243, gnn/evaluate.py:147-240) on a synthetic on-disk
dataset, recorded at the boundary of the hot path and at its very end:

  per batch   the batch tuple (dataset_load.py:623-629), the tensors the encoders hand to the reasoning loop
              (relation features, the initial instructions) - encoders are out of scope -
              and the loop's results: pred_dist, pred, loss;
  once        the state_dict of everything ON the path (type_layer, reasoning, reform0..), the Evaluator's eps,
              entity names, and every line of the ``.info`` file the run wrote.

tests/test_gpu_closed_loop.py replays TypeLayer -> T x (L layers + QueryReform) -> candidate selection on the GPU
with NOTHING recorded fed back in between, and compares with these results.

    python tests/golden/make_golden_e2e.py        ->  tests/golden/rearev_closed_loop.npz
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
sys.path.insert(0, HERE)

import make_golden  # noqa: E402  (dataset writer + reference start-up shims)


def main():
    import parsing
    parsing.create_parser_nutrea = lambda p: None            # reference bug, SURVEY section 4(1)
    from modules.question_encoding import base_encoder
    _orig_init = base_encoder.BaseInstruction.__init__

    def _init(self, args, constraint=False):                  # reference bug, SURVEY section 4(2)
        _orig_init(self, args, constraint)
    base_encoder.BaseInstruction.__init__ = _init

    rng = np.random.default_rng(2026)
    tmp = tempfile.mkdtemp(prefix="gnnrag_gold_e2e_")
    folder = os.path.join(tmp, "synth") + "/"
    make_golden.write_dataset(folder, rng, n_ent=400, n_rel=23, n_q=10)
    D = 50                                                    # the released checkpoints' entity_dim (gnn/README.md:19)
    argv = ["ReaRev", "--data_folder", folder, "--lm", "lstm", "--relation_word_emb", "False",
            "--entity_dim", str(D), "--kg_dim", str(D // 2), "--word_dim", "24",
            "--num_iter", "3", "--num_ins", "2", "--num_gnn", "3", "--batch_size", "4",
            "--test_batch_size", "4", "--checkpoint_dir", tmp + "/", "--experiment_name", "gold",
            "--name", "synth"]
    import argparse
    parser = argparse.ArgumentParser()
    parsing.add_parse_args(parser)
    args = vars(parser.parse_args(argv))
    args["use_cuda"] = False
    args["word_emb_file"] = None
    np.random.seed(args["seed"])
    torch.manual_seed(args["seed"])

    from dataset_load import load_data
    from evaluate import Evaluator
    from models.ReaRev.rearev import ReaRev
    dataset = load_data(args, args["lm"])
    test = dataset["test"]
    model = ReaRev(args, len(dataset["entity2id"]), test.num_kb_relation, dataset["num_word"])
    # a trained model has peaked distributions; random weights give near-uniform ones and empty candidate lists.
    # Sharpen the score so that the candidate selection (threshold + top-p cut) has something to do.
    with torch.no_grad():
        model.reasoning.score_func.weight.mul_(40.0)
    model.eval()

    batches = []
    reasoning = model.reasoning
    orig_init_reason, orig_forward = reasoning.init_reason, reasoning.forward
    cur = {}

    def init_reason_hook(**kw):
        cur.clear()
        cur["init"] = {k: (v.detach().numpy().copy() if torch.is_tensor(v) else v) for k, v in kw.items()}
        cur["calls"] = 0
        return orig_init_reason(**kw)

    def forward_hook(current_dist, relational_ins, step=0, return_score=False):
        if cur["calls"] == 0:                                 # the instructions the encoder produced (rearev.py:192-207)
            cur["ins0"] = relational_ins.detach().numpy().copy()
            cur["seed_dist"] = current_dist.detach().numpy().copy()
        cur["calls"] += 1
        return orig_forward(current_dist, relational_ins, step=step, return_score=return_score)

    reasoning.init_reason, reasoning.forward = init_reason_hook, forward_hook
    orig_model_forward = model.forward

    def model_forward(batch, training=False):
        out = orig_model_forward(batch, training=training)
        loss, pred, pred_dist, _ = out
        init = cur["init"]
        b = dict(local_entity=init["local_entity"], query_entities=init["query_entities"],
                 rel_features=init["rel_features"], rel_features_inv=init["rel_features_inv"],
                 ins0=cur["ins0"], seed_dist=cur["seed_dist"], pred=pred.numpy().copy(),
                 pred_dist=pred_dist.detach().numpy().copy(), loss=float(loss), calls=cur["calls"])
        b.update(make_golden.edge_arrays(init["kb_adj_mat"]))
        batches.append(b)
        return out

    model.forward = model_forward
    ev = Evaluator(args=args, model=model, entity2id=dataset["entity2id"], relation2id=dataset["relation2id"],
                   device=torch.device("cpu"))
    answers = []
    orig_get_batch = test.get_batch

    def get_batch(*a, **kw):
        b = orig_get_batch(*a, **kw)
        answers.append(b[-1])
        return b

    test.get_batch = get_batch
    np.random.seed(77)
    f1, h1, em = ev.evaluate(test, 4, write_info=True)
    info_path = os.path.join(args["checkpoint_dir"], "{}_test.info".format(args["experiment_name"]))
    info_lines = open(info_path).read().splitlines()
    assert len(info_lines) == test.num_data and len(batches) == 3
    n_cand = [len(json.loads(l)["cand"]) for l in info_lines]
    print("f1 %.4f h1 %.4f em %.4f; candidates per question %s" % (f1, h1, em, n_cand))
    assert max(n_cand) >= 2, "the fixture should exercise the top-p cut"

    d = dict(D=D, I=2, L=3, T=3, num_entity=len(dataset["entity2id"]), num_kb_relation=test.num_kb_relation,
             max_local_entity=test.max_local_entity, eps=float(args["eps"]), f1=f1, h1=h1, em=em,
             norm_rel=int(bool(args.get("norm_rel", False))), normalized_gnn=int(bool(args["normalized_gnn"])),
             pos_emb=int(bool(args["pos_emb"])), n_batches=len(batches),
             info=np.array(info_lines), id2entity=np.array([ev.id2entity[i] for i in range(len(ev.id2entity))]))
    for k, b in enumerate(batches):
        for key, v in b.items():
            d["b%d.%s" % (k, key)] = np.asarray(v)
        d["b%d.answers" % k] = np.array(json.dumps([[int(x) for x in a] for a in answers[k]]))
    for prefix, mod in (("type_layer.", model.type_layer), ("reasoning.", reasoning)):
        for k, v in mod.state_dict().items():
            d["param." + prefix + k] = v.numpy().copy()
    j = 0
    while getattr(model, "reform%d" % j, None) is not None:
        for k, v in getattr(model, "reform%d" % j).state_dict().items():
            d["param.reform%d.%s" % (j, k)] = v.numpy().copy()
        j += 1
    out = os.path.join(HERE, "rearev_closed_loop.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
