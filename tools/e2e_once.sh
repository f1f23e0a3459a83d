#!/bin/bash
# one e2e run of the unmodified main.py --is_eval on the staged dataset with the stage timers (no CPU leg):
#   bash tools/e2e_once.sh <variant> <test_batch_size> [ENV=VALUE ...]
V=${1:-d200}; BS=${2:-16}; shift 2
python - "$V" "$BS" "$@" <<'PY'
import json, os, shutil, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import stage_ref
v, bs = sys.argv[1], sys.argv[2]
extra = dict(a.split("=", 1) for a in sys.argv[3:])
argv = list(stage_ref.variant_argv(v)); argv[argv.index("--test_batch_size") + 1] = bs
ck = tempfile.mkdtemp() + "/"
shutil.copyfile(os.path.join(stage_ref.CKPT, stage_ref.ckpt_name(v)), ck + stage_ref.ckpt_name(v))
cmd = [sys.executable, "tools/run_reference.py", stage_ref.GNN] + argv + ["--is_eval", "--load_experiment", stage_ref.ckpt_name(v),
                                                                         "--checkpoint_dir", ck, "--experiment_name", "once"]
r = subprocess.run(cmd, env=dict(os.environ, GNNRAG_DEVICE_FACTS="1", GNNRAG_E2E_TIMES="1", **extra), capture_output=True, text=True)
line = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("GNNRAG_E2E ")]
if r.returncode or not line:
    print((r.stdout + r.stderr)[-2000:]); sys.exit(1)
for c in json.loads(line[-1][11:])["evaluate_calls"]:
    nb = max(c["batches"], 1)
    print("%d questions, %.1f questions/s; per batch: get_batch %.2f ms, structure %.2f, forward %.2f, tail %.2f" % (
        c["questions"], c["questions"] / c["seconds"], 1e3 * c["get_batch_s"] / nb, 1e3 * c["structure_s"] / nb,
        1e3 * (c["forward_s"] - c["structure_s"]) / nb, 1e3 * (c["seconds"] - c["get_batch_s"] - c["forward_s"]) / nb))
PY
