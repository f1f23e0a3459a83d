"""The UNMODIFIED reference entry point on the MI355X ("Next round" item 2; north_star: "drops into
main.py/evaluate.py unchanged ... Hits@1 identical"):

    python tools/run_reference.py oracle/_ref/gnn ReaRev --is_eval --load_experiment synth-final.ckpt ...   (per staged variant)

i.e. gnn/main.py -> Trainer_KBQA -> load_data -> ReaRev (built from the reference's own models/ReaRev/rearev.py, whose
imports resolve to this package's modules) -> load_ckpt of a checkpoint the reference's trainer wrote on CPU ->
Evaluator.evaluate over the valid and test splits -> .info file, with the device-resident fact cache
(GNNRAG_DEVICE_FACTS=1) and the device-side candidate selection - compared with what the PURE reference produced on CPU
for the same checkpoint and data (oracle/stage_ref.py: expected_test.info, expected.json):

  * per question the same candidates in the same order, their probabilities within 1e-4 (north_star's bar),
  * identical per-question precision / recall / F1 / hit / EM and identical logged F1 / H@1 / EM of both splits;

once more with GNNRAG_FORCE_DIST=1, which puts shard_model and RCCL (all-gather of the scored nodes, all-reduce of
the loss) on the path with world size 1, and once with GNNRAG_DEVICE_STRUCTURES=1 (per-question sorted structures cached
on the GPU, a batch's structure = their concatenation).

oracle/_ref (staged reference sources + synthetic dataset + checkpoint + CPU expectations) is git-ignored and built by
``python oracle/stage_ref.py`` / ``__graft_entry__.build()`` in the build container; it travels to the GPU box with the
snapshot.  Without it the test skips.  Reference: gnn/main.py:30-44, gnn/train_model.py:193-198, gnn/evaluate.py:147-240.
"""
import json
import os
import re
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import stage_ref  # noqa: E402  (paths, per-variant argv; test infrastructure)

GNN, CKPT = stage_ref.GNN, stage_ref.CKPT
STAGED = stage_ref.staged()
# __graft_entry__.build() leaves this marker when staging FAILED in the build container: an unstaged tree is then a broken
# parity leg, not an optional one - the tests below fail instead of skipping (GNNRAG_ALLOW_UNSTAGED=1 to skip knowingly)
STAGING_FAILED = os.path.exists(os.path.join(REPO, "oracle", "_ref", "STAGING_FAILED")) and not os.environ.get("GNNRAG_ALLOW_UNSTAGED")


@pytest.mark.gpu
def test_staging_did_not_fail_silently():
    assert not STAGING_FAILED, ("oracle/stage_ref.py failed in the build container (see oracle/_ref/STAGING_FAILED): the "
                                "unmodified-main.py parity tests cannot run; set GNNRAG_ALLOW_UNSTAGED=1 to skip them knowingly")

TOL = 1e-4
TIE = 1e-6          # candidates this close in the reference's own output may come out in either order


def _run_main_py(tmp_path, variant, extra_env, tag):
    ck = str(tmp_path) + "/"
    shutil.copyfile(os.path.join(CKPT, stage_ref.ckpt_name(variant)), os.path.join(ck, stage_ref.ckpt_name(variant)))
    argv = [sys.executable, os.path.join(REPO, "tools", "run_reference.py"), GNN] + stage_ref.variant_argv(variant) + [
        "--is_eval", "--load_experiment", stage_ref.ckpt_name(variant), "--checkpoint_dir", ck, "--experiment_name", "gpu"]
    import socket
    with socket.socket() as sk:                       # a free rendezvous port for the forced process group
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, GNNRAG_DEVICE_FACTS="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **extra_env)
    r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=900)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-4000:]
    metrics = {}
    for key in ("EVAL", "TEST"):
        hit = re.findall(key + r" F1: ([0-9.]+), H1: ([0-9.]+), EM ([0-9.]+)", log)
        assert hit, log[-2000:]
        metrics[key.lower()] = [float(x) for x in hit[-1]]
    lines = open(os.path.join(ck, "gpu_test.info")).read().splitlines()
    out_dir = os.path.join(REPO, "gpurun_out", "main_py")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "run_%s.log" % tag), "w") as f:
        f.write(log[-20000:])
    shutil.copyfile(os.path.join(ck, "gpu_test.info"), os.path.join(out_dir, "gpu_test_%s.info" % tag))
    return metrics, [json.loads(l) for l in lines], log


CASES = [("d50", "single"), ("d50", "force_dist"), ("d50", "structure_cache"), ("d200", "single"), ("d200", "structure_cache"),
         ("cwq", "single"),
         # round 6: a 6000-type relation vocabulary (<= 300 per question: relation compaction, 6002-row tables), the released
         # CWQ flags (--num_iter 2 --num_ins 3, gnn/scripts/rearev_cwq.sh:14), and --normalized_gnn true --pos_emb --norm_rel
         ("fb6k", "single"), ("fb6k", "structure_cache"), ("cwqflags", "single"), ("normpos", "single"),
         # --eps 0.3: ~10 retrieved candidates per question (the d200 checkpoint)
         ("d200eps", "single")]
# variants whose reference checkpoint need not answer a good part of the questions (a configuration the CPU trainer does
# not learn within its budget still has to come out candidate for candidate like the reference's)
NO_H1_GUARD = ("fb6k", "cwqflags", "normpos")       # (8-22 epochs of the reference's CPU trainer leave all three on the
                                                      # "uniform over the seed's neighbourhood" plateau: H@1 0.14-0.18)


@pytest.mark.skipif(not STAGED, reason="oracle/_ref not staged (python oracle/stage_ref.py in the build container)")
@pytest.mark.parametrize("variant,mode", CASES, ids=["%s-%s" % c for c in CASES])
def test_unmodified_main_py_eval_matches_cpu_reference(tmp_path, variant, mode):
    """520 test + 160 dev questions of the LEARNABLE staged dataset (a relation path from the seed determines the answer;
    subgraphs up to 2000 entities, 16 test questions with a hub row of > 4096 facts), checkpoints trained by the
    reference's own trainer on CPU: d50 (released-checkpoint dims), d200 (the benchmark's hidden size), cwq (--name cwq:
    the seed keeps its candidate slot, dataset_load.py:249-257)."""
    if not stage_ref.staged_variant(variant):
        pytest.skip("variant %s not staged (python oracle/stage_ref.py)" % variant)
    want_metrics = json.load(open(os.path.join(CKPT, "expected_%s.json" % variant)))
    want = [json.loads(l) for l in open(os.path.join(CKPT, "expected_%s_test.info" % variant)).read().splitlines()]
    # the comparison is only worth something when the reference itself answers a good part of the questions
    if variant not in NO_H1_GUARD:
        assert 0.3 <= want_metrics["test"][1] <= 0.9 and 0.3 <= want_metrics["eval"][1] <= 0.95, want_metrics
    assert len(want) >= 500
    env = {"force_dist": {"GNNRAG_FORCE_DIST": "1"}, "structure_cache": {"GNNRAG_DEVICE_STRUCTURES": "1"}}.get(mode, {})
    metrics, got, log = _run_main_py(tmp_path, variant, env, "%s_%s" % (variant, mode))
    assert "gnnrag_amd: native library mapped" in log, log[-2000:]        # the child ran on libgnnrag_hip.so
    for split in ("eval", "test"):                                        # logged with 4 decimals by the reference
        assert metrics[split] == want_metrics[split], (split, metrics, want_metrics)
    assert len(got) == len(want) and len(got) > 0
    worst, reordered, cut_ties = 0.0, 0, 0
    for g, w in zip(got, want):
        assert g["question"] == w["question"] and g["answers"] == w["answers"]
        for key in ("precison", "recall", "f1", "hit", "em"):
            assert g[key] == w[key], (key, g["question"])
        # the same retrieved entities, in the same order - except that two candidates whose reference probabilities
        # differ by less than TIE (1e-6: two orders of magnitude below north_star's 1e-4 bar) may be swapped: a
        # question the model is unsure about retrieves up to 1800 candidates of nearly equal probability, and fp32
        # sums in a different order decide such ties differently
        ge, we = [c[0] for c in g["cand"]], [c[0] for c in w["cand"]]
        pw_of, pg_of = dict(map(tuple, w["cand"])), dict(map(tuple, g["cand"]))
        assert len(ge) == len(we), g["question"]
        if set(ge) != set(we):
            # the top-p cut fell inside a group of (near-)equal probabilities (structurally equivalent nodes): which
            # members of the group are retrieved depends on the order among ties - every entity only one side retrieved
            # must sit within TIE of the cut, i.e. of the smallest retrieved probability
            cut = min(pw_of.values())
            for e in set(ge) ^ set(we):
                assert abs((pg_of[e] if e in pg_of else pw_of[e]) - cut) <= TIE, (g["question"], e)
            cut_ties += 1
        if ge != we:
            reordered += 1
            prob = lambda e: pw_of[e] if e in pw_of else pg_of[e]
            for a, b in zip(ge, we):
                assert a == b or abs(prob(a) - prob(b)) <= TIE, (g["question"], a, b, prob(a), prob(b))
        worst = max([worst] + [abs(pg_of[e] - pw_of[e]) for e in we if e in pg_of])
    assert worst <= TOL, worst
    assert reordered <= len(got) // 10 and cut_ties <= 3, (reordered, cut_ties)       # ties are the exception
    print("main.py on the MI355X (%s, %s): %d questions, H@1 %.4f (dev %.4f), max |candidate probability - CPU reference| = %.3g, "
          "%d questions with a swap among near-equal candidates (|dp| <= %.0e), %d with the top-p cut inside such a group"
          % (variant, mode, len(got), metrics["test"][1], metrics["eval"][1], worst, reordered, TIE, cut_ties))
