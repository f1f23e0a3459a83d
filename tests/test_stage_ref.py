"""oracle/stage_ref.py (build container only): the recipe that lets the unmodified reference entry point run on the GPU
box stages the reference sources, a synthetic dataset in the reference's on-disk format, a checkpoint written by the
reference's own trainer and the CPU reference's evaluation of it into the git-ignored oracle/_ref/.  Checked here: the
staged pieces exist after a build, the expectations parse, and nothing of it is tracked by git."""
import json
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/gnn"


@pytest.mark.skipif(not os.path.isdir(REF), reason="live reference not available")
def test_staged_reference_is_complete_and_untracked():
    import sys
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stage_ref
    if not stage_ref.staged():
        stage_ref.main()
    assert stage_ref.staged()
    # round-6 variants whose checkpoints stay on the reference trainer's plateau (flag coverage, not accuracy: VERDICT round 5)
    unconverged = ("fb6k", "cwqflags", "normpos")
    for v in stage_ref.VARIANTS:
        if v in stage_ref.ROUND6_VARIANTS and not stage_ref.staged_variant(v):
            continue                             # staged one by one: a missing one skips its GPU case, nothing else
        exp = json.load(open(os.path.join(stage_ref.CKPT, "expected_%s.json" % v)))
        assert {"eval", "test"} <= set(exp) and all(len(exp[k]) == 3 for k in ("eval", "test"))
        # a LEARNABLE dataset: the CPU reference answers a good part of the questions (it was 0 / 48)
        assert v in unconverged or 0.3 <= exp["test"][1] <= 0.9, (v, exp)
        lines = open(os.path.join(stage_ref.CKPT, "expected_%s_test.info" % v)).read().splitlines()
        assert len(lines) >= 500 and all("cand" in json.loads(l) for l in lines)
    # the test split holds subgraphs of WebQSP's padded width with a hub row of more than 4096 facts
    import collections
    big = 0
    for line in open(os.path.join(stage_ref.DATA, "test.json")):
        q = json.loads(line)
        if len(q["subgraph"]["entities"]) >= 1700:
            indeg = collections.Counter(t[2] for t in q["subgraph"]["tuples"])
            big += max(indeg.values()) > 4096
    assert big >= 8
    # the staged sources are a verbatim copy of the reference's files (never edited) ...
    for rel in ("main.py", "evaluate.py", "models/ReaRev/rearev.py", "modules/kg_reasoning/reasongnn.py"):
        assert open(os.path.join(stage_ref.GNN, rel), "rb").read() == open(os.path.join(REF, rel), "rb").read()
    # ... and stay out of the history
    tracked = subprocess.run(["git", "-C", REPO, "ls-files", "oracle/_ref"], capture_output=True, text=True).stdout.strip()
    assert tracked == ""
    assert "oracle/_ref/" in open(os.path.join(REPO, ".gitignore")).read()
    gi = os.path.join(REPO, ".gpurunignore")
    assert not os.path.exists(gi) or "oracle/_ref" not in open(gi).read()      # it must travel to the GPU box
