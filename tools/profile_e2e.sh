#!/bin/bash
# host-side profile of the unmodified main.py --is_eval on the MI355X (where does a batch's wall time go?)
# usage: bash tools/profile_e2e.sh <variant> <test_batch_size> [ENV=VALUE ...]
V=${1:-d200}; BS=${2:-16}; shift 2
mkdir -p gpurun_out/e2e_prof
python - "$V" "$BS" "$@" <<'PY'
import os, sys, subprocess, shutil, tempfile
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import stage_ref
v, bs = sys.argv[1], sys.argv[2]
extra = dict(a.split("=", 1) for a in sys.argv[3:])
argv = list(stage_ref.variant_argv(v)); argv[argv.index("--test_batch_size") + 1] = bs
ck = tempfile.mkdtemp() + "/"
shutil.copyfile(os.path.join(stage_ref.CKPT, stage_ref.ckpt_name(v)), ck + stage_ref.ckpt_name(v))
out = "gpurun_out/e2e_prof/%s_b%s%s.prof" % (v, bs, "_" + "_".join(extra) if extra else "")
cmd = [sys.executable, "-m", "cProfile", "-o", out, "tools/run_reference.py", stage_ref.GNN] + argv + [
    "--is_eval", "--load_experiment", stage_ref.ckpt_name(v), "--checkpoint_dir", ck, "--experiment_name", "prof"]
env = dict(os.environ, GNNRAG_DEVICE_FACTS="1", GNNRAG_E2E_TIMES="1", **extra)
r = subprocess.run(cmd, env=env, capture_output=True, text=True)
print([l for l in (r.stdout + r.stderr).splitlines() if l.startswith("GNNRAG_E2E")][-1][:600] if r.returncode == 0 else (r.stdout + r.stderr)[-2000:])
import pstats
st = pstats.Stats(out); st.sort_stats("cumulative")
import io
buf = io.StringIO(); st.stream = buf; st.print_stats(70)
txt = buf.getvalue()
st2 = pstats.Stats(out); st2.sort_stats("tottime")
buf2 = io.StringIO(); st2.stream = buf2; st2.print_stats(60)
txt = txt + "\n\n==== by own time ====\n" + buf2.getvalue()
open(out.replace(".prof", ".txt"), "w").write(txt)
print("\n".join(l[:150] for l in txt.splitlines()[:90]))
PY
find gpurun_out/e2e_prof -name '*.prof' -delete
