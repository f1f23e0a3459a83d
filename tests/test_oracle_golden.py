"""Pins both CPU oracles against fixtures recorded from the LIVE reference modules
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

import oracle.rearev_torch_cpu as otorch
import oracle.rearev_np64 as onp


@pytest.mark.parametrize("name", ["layer_d200.npz", "layer_d50.npz"])
def test_torch_restatement_matches_reference(name):
    cfg, batch, feats, params, ref = load_golden(name)
    out = otorch.run_stack(batch, feats, params)
    for c in range(cfg.T * cfg.L):
        # same library, same op order: only COO summation order inside sparse.mm can differ
        np.testing.assert_allclose(out["h"][c], ref["h"][c], rtol=0, atol=2e-6)
        np.testing.assert_allclose(out["dist"][c], ref["dist"][c], rtol=0, atol=1e-7)
        np.testing.assert_allclose(out["score"][c], ref["score"][c], rtol=1e-6, atol=2e-6)
        assert (out["dist"][c].argmax(1) == ref["dist"][c].argmax(1)).all()


@pytest.mark.parametrize("name", ["layer_d200.npz", "layer_d50.npz"])
def test_factored_np64_matches_reference(name):
    cfg, batch, feats, params, ref = load_golden(name)
    out = onp.run_stack(batch, feats, params, dtype=np.float64)
    for c in range(cfg.T * cfg.L):
        # fp64 factored form vs fp32 reference: fp32 rounding of the reference only
        np.testing.assert_allclose(out["h"][c], ref["h"][c], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out["dist"][c], ref["dist"][c], rtol=0, atol=1e-6)
        valid = ref["score"][c] > -1e10
        np.testing.assert_allclose(out["score"][c][valid], ref["score"][c][valid], rtol=0, atol=2e-5)
        assert (out["dist"][c].argmax(1) == ref["dist"][c].argmax(1)).all()


def test_masked_slots_and_all_masked_question():
    """Masked slots get probability exactly 0; a question whose slots are ALL masked (1 real node =
    the masked seed) softmaxes to exactly uniform 1/N because score-1e11 rounds to -1e11 in fp32."""
    cfg, batch, feats, params, ref = load_golden("layer_d50.npz")
    mask = batch.local_entity != batch.num_entity
    some = mask.any(axis=1)
    assert (~some).any(), "fixture must contain an all-masked question"
    out = otorch.run_stack(batch, feats, params)
    for dist in list(ref["dist"]) + [out["dist"][-1]]:
        assert (dist[some][~mask[some]] == 0).all()
        np.testing.assert_array_equal(dist[~some], np.float32(1.0) / np.float32(cfg.N))


@pytest.mark.parametrize("norm_rel", [False, True])
def test_type_layer_oracles_match_reference(norm_rel):
    z = np.load(os.path.join(GOLDEN, "typelayer.npz"))
    F = len(z["heads"])
    et = (z["heads"], z["rels"], z["tails"], z["batch_ids"], np.arange(F), z["weight_list"].tolist(),
          z["weight_rel_list"].tolist())
    B, N = int(z["B"]), int(z["N"])
    W = z["param.type_layer.kb_self_linear.weight"]
    b = z["param.type_layer.kb_self_linear.bias"]
    ref = z["ref.h0_norm%d" % int(norm_rel)]
    import torch
    got_t = otorch.type_layer(et, B, N, torch.from_numpy(z["feat.rel_features"]), torch.from_numpy(W),
                              torch.from_numpy(b), norm_rel).numpy()
    np.testing.assert_allclose(got_t, ref, rtol=0, atol=2e-6)
    got_n = onp.type_layer(et, B, N, z["feat.rel_features"], W, b, norm_rel)
    np.testing.assert_allclose(got_n, ref, rtol=0, atol=2e-5)


def test_e2e_call_site_fixture_replays_through_oracle():
    """rearev_e2e.npz holds the tensors crossing the layer boundary inside a real
    ReaRev.forward (dataset_load -> get_batch -> model).  Replaying the recorded calls
    through the oracle must reproduce the recorded outputs, including the final pred."""
    import torch
    z = np.load(os.path.join(GOLDEN, "rearev_e2e.npz"))
    B, N = int(z["B"]), int(z["N"])
    F = len(z["heads"])
    et = (z["heads"], z["rels"], z["tails"], z["batch_ids"], np.arange(F), z["weight_list"].tolist(),
          z["weight_rel_list"].tolist())
    st = otorch.Structure(et, B, N, normalized_gnn=False)
    params = otorch.to_torch_params({k[6:]: z[k] for k in z.files if k.startswith("param.")})
    mask = torch.from_numpy((z["local_entity"] != int(z["num_entity"])).astype(np.float32))
    h = torch.from_numpy(z["h0"])
    rf, rfi = torch.from_numpy(z["rel_features"]), torch.from_numpy(z["rel_features_inv"])
    for c, step in enumerate(z["call.step"]):
        _, dist, h = otorch.layer_forward(st, h, mask, torch.from_numpy(z["call.dist_in"][c]),
                                          torch.from_numpy(z["call.ins"][c]), params, int(step), rf, rfi, False)
        np.testing.assert_allclose(dist.numpy(), z["call.dist_out"][c], rtol=0, atol=1e-7)
        np.testing.assert_allclose(h.numpy(), z["call.h_out"][c], rtol=0, atol=2e-6)
    np.testing.assert_allclose(dist.numpy(), z["pred_dist"], rtol=0, atol=1e-7)
    assert (dist.numpy().argmax(1) == z["pred"]).all()


@pytest.mark.parametrize("name", ["layer_d200.npz", "layer_d50.npz"])
def test_grad_oracle_matches_reference_autograd(name):
    """The float64 autograd restatement (oracle/rearev_grad.py) against gradients recorded from the
    live reference modules (tests/golden/make_golden_grad.py)."""
    import oracle.rearev_grad as og
    cfg, batch, feats, params, _ = load_golden(name)
    z = np.load(os.path.join(GOLDEN, "grad_" + name))
    got = og.stack_grads(batch, feats, params, z["cot.Gd"], z["cot.Gh"])
    assert abs(got["loss"] - float(z["loss"])) <= 1e-4 * max(1.0, abs(float(z["loss"])))
    names = [k[5:] for k in z.files if k.startswith("grad.")]
    assert len(names) >= 12
    for k in names:
        want = z["grad." + k]
        tol = 2e-4 * max(np.abs(want).max(), 1e-3)           # the reference ran in fp32
        np.testing.assert_allclose(got[k], want, rtol=0, atol=tol, err_msg=k)


def test_typelayer_grad_oracle_matches_reference_autograd():
    import oracle.rearev_grad as og
    z = np.load(os.path.join(GOLDEN, "typelayer.npz"))
    zg = np.load(os.path.join(GOLDEN, "grad_typelayer.npz"))
    B, N, D = int(z["B"]), int(z["N"]), int(z["D"])
    et = (z["heads"], z["rels"], z["tails"])
    W = z["param.type_layer.kb_self_linear.weight"].astype(np.float64)
    b = z["param.type_layer.kb_self_linear.bias"].astype(np.float64)
    rf = z["feat.rel_features"].astype(np.float64)
    T = rf @ W.T + b
    for norm_rel in (False, True):
        h0 = z["ref.h0_norm%d" % int(norm_rel)].reshape(B * N, D)
        g_pre = zg["cot.G"].reshape(B * N, D) * (h0 > 0)
        g_T = og.typelayer_grad(et, B, N, T, g_pre, z["weight_rel_list"] if norm_rel else None)
        tag = "grad.norm%d." % int(norm_rel)
        for name, got in (("rel_features", g_T @ W), ("kb_self_linear.weight", g_T.T @ rf),
                          ("kb_self_linear.bias", g_T.sum(0))):
            want = zg[tag + name]
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-4 * max(np.abs(want).max(), 1e-3), err_msg=name)


def test_eval_tail_restatement_reproduces_the_reference_info_file():
    """oracle/eval_tail.py (candidate filter, stable sort, top-p cut, f1_and_hits, the .info record) fed with the
    pred_dist the LIVE reference computed reproduces every line the reference's Evaluator wrote, byte for byte
    (tests/golden/rearev_closed_loop.npz, made by make_golden_e2e.py)."""
    import json
    import oracle.eval_tail as oe
    z = np.load(os.path.join(GOLDEN, "rearev_closed_loop.npz"))
    eps, N, T = float(z["eps"]), int(z["max_local_entity"]), int(z["T"])
    ignore_prob = (1 - eps) / N
    id2entity = {i: str(s) for i, s in enumerate(z["id2entity"])}
    pad = len(id2entity)
    lines = [str(l) for l in z["info"]]
    qi = 0
    for k in range(int(z["n_batches"])):
        g = lambda name: z["b%d.%s" % (k, name)]
        answers = json.loads(str(g("answers")))
        probs, cands, seeds = g("pred_dist"), g("local_entity"), g("query_entities")
        for b in range(probs.shape[0]):
            kept, cut = oe.select(probs[b].tolist(), cands[b].tolist(), seeds[b].tolist(), pad, ignore_prob, eps)
            # the reference hands f1_and_hits EVERY kept candidate; it cuts the list itself
            cand2prob = [(int(cands[b, j]), float(probs[b, j])) for j in kept]
            question = json.loads(lines[qi])["question"]
            rec = oe.info_record(question, T, answers[b], cand2prob, id2entity, None, eps)
            assert json.dumps(rec) == lines[qi]
            assert len(rec["cand"]) == cut
            qi += 1
    assert qi == len(lines)


def test_oracle_closed_loop_reproduces_the_reference_forward():
    """The restatements chained the way ReaRev.forward chains the real modules (TypeLayer -> T x (L layers +
    QueryReform), rearev.py:163-243), started from the encoder outputs the live reference recorded, reproduce its
    pred_dist and pred on every batch of the closed-loop fixture - the chain the GPU closed-loop test relies on is
    pinned end to end on the CPU as well."""
    import torch
    import oracle.query_update_torch as oq
    z = np.load(os.path.join(GOLDEN, "rearev_closed_loop.npz"))
    D, I, L, T = (int(z[k]) for k in ("D", "I", "L", "T"))
    p = {k[len("param.reasoning."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.reasoning.")}
    tw, tb = (torch.from_numpy(z["param.type_layer.kb_self_linear." + s]) for s in ("weight", "bias"))
    rf = [{s: torch.from_numpy(z["param.reform%d.%s" % (j, s)]) for s in
           ("q_ent_attn.weight", "q_ent_attn.bias", "fusion.r.weight", "fusion.g.weight")} for j in range(I)]
    num_entity = int(z["num_entity"])
    for k in range(int(z["n_batches"])):
        g = lambda name: z["b%d.%s" % (k, name)]
        F = len(g("heads"))
        et = (g("heads"), g("rels"), g("tails"), g("batch_ids"), np.arange(F), g("weight_list").tolist(),
              g("weight_rel_list").tolist())
        local_entity = torch.from_numpy(g("local_entity"))
        B, N = local_entity.shape
        relfeat, relfeat_inv = torch.from_numpy(g("rel_features")), torch.from_numpy(g("rel_features_inv"))
        qe = torch.from_numpy(g("query_entities")).float()
        seed = torch.from_numpy(g("seed_dist")).float()
        mask = (local_entity != num_entity).float()
        h = otorch.type_layer(et, B, N, relfeat, tw, tb, bool(int(z["norm_rel"])))
        st = otorch.Structure(et, B, N, bool(int(z["normalized_gnn"])))
        ins = [torch.from_numpy(g("ins0"))[:, j] for j in range(I)]
        for _ in range(T):
            rel_ins = torch.stack(ins, dim=1)
            dist = seed
            for j in range(L):
                _, dist, h = otorch.layer_forward(st, h, mask, dist, rel_ins, p, j, relfeat, relfeat_inv,
                                                  bool(int(z["pos_emb"])))
            ins = [oq.query_reform(ins[j], h, qe, local_entity, rf[j]["q_ent_attn.weight"], rf[j]["q_ent_attn.bias"],
                                   rf[j]["fusion.r.weight"], rf[j]["fusion.g.weight"]) for j in range(I)]
        assert np.abs(dist.numpy() - g("pred_dist")).max() <= 1e-6
        assert np.array_equal(dist.argmax(1).numpy(), g("pred"))


def test_lstm_oracle_matches_the_live_reference_encoder():
    """oracle/lstm_np64.py against what the LIVE reference's LSTMInstruction.encode_question produced
    (tests/golden/lstm_encoder.npz, tests/golden/make_golden_lstm.py): hidden states of every token and the final
    (h_n, c_n), ragged questions padded with the pad word; fp32 reference vs float64 restatement."""
    import oracle.lstm_np64 as lstm_np64
    g = np.load(os.path.join(GOLDEN, "lstm_encoder.npz"))
    for tag in ("d50", "d128"):
        P = {k.split(".param.")[1]: g[k] for k in g.files if k.startswith(tag + ".param.")}
        out, h, c = lstm_np64.lstm_forward(g[tag + ".word_emb"], P["node_encoder.weight_ih_l0"], P["node_encoder.weight_hh_l0"],
                                           P["node_encoder.bias_ih_l0"], P["node_encoder.bias_hh_l0"])
        assert np.abs(out - g[tag + ".query_hidden_emb"]).max() <= 2e-6
        assert np.abs(h - g[tag + ".h_n"]).max() <= 2e-6 and np.abs(c - g[tag + ".c_n"]).max() <= 4e-6
        assert np.array_equal(g[tag + ".query_node_emb"][:, 0], g[tag + ".h_n"])           # lstm_encoder.py:40
