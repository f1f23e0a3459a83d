#!/bin/bash
# where the HOST time of a steady-state ReaRev.forward goes (unmodified main.py --is_eval on the MI355X):
#   bash tools/profile_forward.sh <variant> <test_batch_size> <torch|cprofile> [ENV=VALUE ...]  ->  gpurun_out/fwd_prof/*.txt
V=${1:-d200}; BS=${2:-64}; KIND=${3:-torch}; shift 3
mkdir -p gpurun_out/fwd_prof
python - "$V" "$BS" "$KIND" "$@" <<'PY'
import os, sys, subprocess, shutil, tempfile
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import stage_ref
v, bs, kind = sys.argv[1], sys.argv[2], sys.argv[3]
extra = dict(a.split("=", 1) for a in sys.argv[4:])
argv = list(stage_ref.variant_argv(v)); argv[argv.index("--test_batch_size") + 1] = bs
ck = tempfile.mkdtemp() + "/"
shutil.copyfile(os.path.join(stage_ref.CKPT, stage_ref.ckpt_name(v)), ck + stage_ref.ckpt_name(v))
tag = "%s_b%s_%s%s" % (v, bs, kind, "_" + "_".join("%s%s" % kv for kv in extra.items()) if extra else "")
out = "gpurun_out/fwd_prof/%s.txt" % tag
cmd = [sys.executable, "tools/run_reference.py", stage_ref.GNN] + argv + [
    "--is_eval", "--load_experiment", stage_ref.ckpt_name(v), "--checkpoint_dir", ck, "--experiment_name", "prof"]
env = dict(os.environ, GNNRAG_DEVICE_FACTS="1", GNNRAG_E2E_TIMES="1", GNNRAG_PROFILE_FORWARD=out, GNNRAG_PROFILE_KIND=kind)
env.update(extra)
r = subprocess.run(cmd, env=env, capture_output=True, text=True)
print(tag, [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("GNNRAG_E2E")][-1][:400] if r.returncode == 0 else (r.stdout + r.stderr)[-2000:])
print("\n".join(l[:160] for l in open(out).read().splitlines()[:40]) if os.path.exists(out) else "no profile written")
PY
