#!/usr/bin/env python
"""TEST INFRASTRUCTURE (checker only; nothing under gnn-rag_amd/ may import this).

Recipe that lets the UNMODIFIED reference entry point (``gnn/main.py`` -> ``Trainer_KBQA.evaluate_single`` ->
``Evaluator.evaluate`` -> ``ReaRev.forward``; gnn/main.py:30-44, gnn/train_model.py:193-198, gnn/evaluate.py:147-240,
gnn/models/ReaRev/rearev.py:163-243) run on the GPU box, where /root/reference does not exist:

  oracle/_ref/gnn/           the reference's gnn/ sources, STAGED from /root/reference/gnn where they lie (git-ignored:
                             never part of the history; travels to the GPU box with the gpurun snapshot like a built .so)
  oracle/_ref/data/synth/    synthetic on-disk dataset in the reference's own format (entities.txt, relations.txt,
                             vocab.txt, {train,dev,test}.json, word_emb.npy - dataset_load.py:45-55,228-238,565-575)
  oracle/_ref/ckpt/          a checkpoint written by the reference's own Trainer_KBQA on CPU (a few train_epoch calls,
                             save_ckpt) + the CPU reference's own evaluation of it through main.py --is_eval:
                             ``expected_test.info`` (per-question candidates + probabilities) and ``expected.json``
                             (F1 / H@1 / EM of the valid and test splits as the reference logged them)

tests/test_gpu_main_py.py then runs ``tools/run_reference.py oracle/_ref/gnn ReaRev --is_eval ...`` on the MI355X and
compares its .info file and metrics with these.  Run here (build container):  python oracle/stage_ref.py
"""
import json
import os
import re
import runpy
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = "/root/reference/gnn"
DST = os.path.join(HERE, "_ref")
GNN = os.path.join(DST, "gnn")
DATA = os.path.join(DST, "data", "synth") + "/"
CKPT = os.path.join(DST, "ckpt") + "/"
EXP = "synth"

# main.py's own flags for this dataset (released-checkpoint dims: gnn/README.md:19)
MODEL_ARGV = ["ReaRev", "--data_folder", DATA, "--lm", "lstm", "--relation_word_emb", "False",
              "--entity_dim", "50", "--kg_dim", "25", "--num_iter", "3", "--num_ins", "2", "--num_gnn", "3",
              "--batch_size", "16", "--test_batch_size", "16", "--name", "synth"]


def shim_reference_startup_bugs():
    """The reference cannot start as shipped: ``parsing.add_parse_args`` calls an undefined
    ``create_parser_nutrea`` and ``LSTMInstruction`` does not pass ``constraint`` to its base class.  Same two shims
    as tests/golden/make_golden.py and tools/run_reference.py; nothing else of the reference is touched."""
    import parsing
    if not hasattr(parsing, "create_parser_nutrea"):
        parsing.create_parser_nutrea = lambda p: None
    from modules.question_encoding import base_encoder
    if not getattr(base_encoder.BaseInstruction.__init__, "_gnnrag_shim", False):
        orig = base_encoder.BaseInstruction.__init__

        def _init(self, args, constraint=False):
            orig(self, args, constraint)
        _init._gnnrag_shim = True
        base_encoder.BaseInstruction.__init__ = _init


def write_dataset(folder, seed=314, n_ent=6000, n_rel=48, n_q=48, n_min=40, n_max=400):
    """Freebase-shaped question subgraphs on disk: per question a seed entity, 40..400 subgraph entities, 3-6 typed
    edges per entity with Zipf-distributed heads (hubs), one or two answers among the seed's 2-hop neighbourhood."""
    import numpy as np
    rng = np.random.default_rng(seed)
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "entities.txt"), "w") as f:
        for i in range(n_ent):
            f.write("m.%05d\n" % i)
    with open(os.path.join(folder, "relations.txt"), "w") as f:
        for i in range(n_rel):
            f.write("dom%d.type%d.rel%d\n" % (i % 5, i % 11, i))
    words = ["what", "is", "the", "name", "of", "who", "where", "film", "city", "born", "wrote", "plays", "in", "team"]
    with open(os.path.join(folder, "vocab.txt"), "w") as f:
        for w in words:
            f.write(w + "\n")
    np.save(os.path.join(folder, "word_emb.npy"), (0.3 * rng.standard_normal((len(words), 24))).astype(np.float32))
    for split in ("train", "dev", "test"):
        with open(os.path.join(folder, split + ".json"), "w") as f:
            for qi in range(n_q):
                n_sub = int(rng.integers(n_min, n_max + 1))
                ents = rng.choice(n_ent, size=n_sub, replace=False)
                n_edge = int(rng.integers(3 * n_sub, 6 * n_sub))
                h = ents[(rng.zipf(1.6, size=n_edge) - 1) % n_sub]
                t = ents[rng.integers(0, n_sub, size=n_edge)]
                r = rng.integers(0, n_rel, size=n_edge)
                tuples = [[int(a), int(b), int(c)] for a, b, c in zip(h, r, t)]
                seed_e = int(ents[0])                             # the hub: most facts start here
                hop1 = {int(c) for a, _, c in tuples if a == seed_e and c != seed_e}
                pool = sorted(hop1) or [int(e) for e in ents[1:]]
                ans = [pool[int(i)] for i in rng.choice(len(pool), size=min(len(pool), int(rng.integers(1, 3))), replace=False)]
                q = " ".join(rng.choice(words, size=int(rng.integers(3, 8))).tolist())
                f.write(json.dumps({
                    "id": "%s-%d" % (split, qi), "question": q, "entities": [seed_e],
                    "answers": [{"kb_id": "m.%05d" % a, "text": "a"} for a in ans],
                    "subgraph": {"tuples": tuples, "entities": [int(e) for e in ents]}}) + "\n")


def stage_sources():
    if not os.path.isdir(SRC):
        raise SystemExit("oracle/stage_ref.py: %s not present (the GPU box uses the files staged in the build container)" % SRC)
    if os.path.isdir(GNN):
        shutil.rmtree(GNN)
    shutil.copytree(SRC, GNN, ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))


def make_checkpoint():
    """The reference's own trainer on CPU: three epochs over the synthetic train split, then save_ckpt('final').
    The synthetic relations carry no signal, so the trained distributions stay near-uniform; a trained model's are peaked.
    score_func is therefore sharpened (x30, as tests/golden/make_golden_e2e.py does) so that the Evaluator's threshold and
    top-p cut have something to decide."""
    import argparse
    import numpy as np
    import torch
    sys.path.insert(0, GNN)
    shim_reference_startup_bugs()
    import parsing
    parser = argparse.ArgumentParser()
    parsing.add_parse_args(parser)
    args = parser.parse_args(MODEL_ARGV + ["--checkpoint_dir", CKPT, "--experiment_name", EXP, "--lr", "0.005"])
    args.use_cuda = False
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    os.makedirs(CKPT, exist_ok=True)
    from train_model import Trainer_KBQA
    from utils import create_logger
    trainer = Trainer_KBQA(args=vars(args), model_name=args.model_name, logger=create_logger(args))
    for epoch in range(3):
        loss, _, h1, f1 = trainer.train_epoch()
        print("stage_ref: epoch %d loss %.4f train h1 %.3f f1 %.3f" % (epoch + 1, loss, np.mean(h1), np.mean(f1)))
    with torch.no_grad():
        trainer.model.reasoning.score_func.weight.mul_(30.0)
    trainer.save_ckpt("final")


def reference_eval_cpu():
    """main.py --is_eval of the pure reference on CPU (own process: main.py parses sys.argv and builds loggers)."""
    code = ("import sys, runpy; sys.path.insert(0, %r); sys.path.insert(0, %r); import stage_ref; "
            "stage_ref.shim_reference_startup_bugs(); sys.argv = ['main.py'] + %r; "
            "runpy.run_path(%r, run_name='__main__')") % (
        GNN, HERE, MODEL_ARGV + ["--is_eval", "--load_experiment", EXP + "-final.ckpt", "--checkpoint_dir", CKPT,
                                 "--experiment_name", EXP + "_cpu"], os.path.join(GNN, "main.py"))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", code], cwd=GNN, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("reference evaluation failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    out = r.stdout + r.stderr
    metrics = parse_metrics(out)
    shutil.copyfile(os.path.join(CKPT, EXP + "_cpu_test.info"), os.path.join(CKPT, "expected_test.info"))
    with open(os.path.join(CKPT, "expected.json"), "w") as f:
        json.dump(metrics, f, indent=1)
    return metrics


def parse_metrics(log_text):
    """The two lines Trainer_KBQA.evaluate_single logs (train_model.py:195-198)."""
    m = {}
    for key in ("EVAL", "TEST"):
        hit = re.findall(key + r" F1: ([0-9.]+), H1: ([0-9.]+), EM ([0-9.]+)", log_text)
        if not hit:
            raise SystemExit("no %s metrics line in the reference's log:\n%s" % (key, log_text[-2000:]))
        m[key.lower()] = [float(x) for x in hit[-1]]
    return m


def staged() -> bool:
    return all(os.path.exists(p) for p in (os.path.join(GNN, "main.py"), os.path.join(DATA, "test.json"),
                                           os.path.join(CKPT, EXP + "-final.ckpt"), os.path.join(CKPT, "expected_test.info"),
                                           os.path.join(CKPT, "expected.json")))


def main(force=False):
    if staged() and not force:
        print("oracle/_ref already staged")
        return
    stage_sources()
    write_dataset(DATA)
    # own process: the reference's modules must not leak into the caller's sys.modules / logging setup
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import stage_ref; stage_ref.make_checkpoint()" % HERE],
                       cwd=GNN, env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""), capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("checkpoint creation failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("stage_ref:")))
    metrics = reference_eval_cpu()
    lines = open(os.path.join(CKPT, "expected_test.info")).read().splitlines()
    ncand = [len(json.loads(l)["cand"]) for l in lines]
    print("oracle/_ref staged: CPU reference metrics %s; candidates per test question: min %d max %d" %
          (metrics, min(ncand), max(ncand)))


if __name__ == "__main__":
    main(force="--force" in sys.argv)
