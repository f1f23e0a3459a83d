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
    exp = json.load(open(os.path.join(stage_ref.CKPT, "expected.json")))
    assert set(exp) == {"eval", "test"} and all(len(v) == 3 for v in exp.values())
    lines = open(os.path.join(stage_ref.CKPT, "expected_test.info")).read().splitlines()
    assert len(lines) == 48 and all("cand" in json.loads(l) for l in lines)
    # the staged sources are a verbatim copy of the reference's files (never edited) ...
    for rel in ("main.py", "evaluate.py", "models/ReaRev/rearev.py", "modules/kg_reasoning/reasongnn.py"):
        assert open(os.path.join(stage_ref.GNN, rel), "rb").read() == open(os.path.join(REF, rel), "rb").read()
    # ... and stay out of the history
    tracked = subprocess.run(["git", "-C", REPO, "ls-files", "oracle/_ref"], capture_output=True, text=True).stdout.strip()
    assert tracked == ""
    assert "oracle/_ref/" in open(os.path.join(REPO, ".gitignore")).read()
    gi = os.path.join(REPO, ".gpurunignore")
    assert not os.path.exists(gi) or "oracle/_ref" not in open(gi).read()      # it must travel to the GPU box
