#!/usr/bin/env python
"""Wall time of ONE training step as the reference's trainer runs it (``Trainer_KBQA.train_epoch``,
gnn/train_model.py:209-233): ``get_batch`` -> ``zero_grad`` -> ``model(batch, training=True)`` -> ``loss.backward()`` ->
``clip_grad_norm_`` -> ``Adam.step()`` -> ``loss.item()``, on the staged synthetic dataset (oracle/stage_ref.py) with the
reference's OWN trainer object, model code, optimiser and loader:

    python tools/time_train_step.py oracle/_ref/gnn --variant d200 [--pure] [--steps 8] [--warm 3] [main.py flags]

  * default: this package's modules underneath (``install.install()`` before the reference imports its models; the train
    split's ``_build_fact_mat`` vectorised with the reference's RNG stream kept) on the MI355X - forward through the
    autograd form of the layers, the typed-edge aggregation's backward in HIP (``aggregate_bwd.hip``);
  * ``--pure``: the reference as it is on the host cores (CUDA_VISIBLE_DEVICES="" by the caller), the baseline.

The trainer starts from the variant's staged checkpoint (realistic weights), seeds numpy / torch with the reference's
seed, and walks the train split in the reference's own shuffled order.  Prints one ``GNNRAG_TRAIN {json}`` line: median /
min / max step time, the split by stage (seams synchronised with the device) and the loss of every step.  bench.py's
``train_step`` block runs both modes as subprocesses."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _limit_pools(pure):
    if pure or os.environ.get("GNNRAG_HOST_THREADS") == "0":
        return
    budget = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            budget = min(budget, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    n = str(max(1, min(8, budget // 2)))
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(k, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("gnn_dir")
    ap.add_argument("--variant", default="d200")
    ap.add_argument("--pure", action="store_true")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    # anything else is handed to the reference's own parser (main.py's flags), e.g. --linear_dropout 0 --lm_dropout 0
    a, extra = ap.parse_known_args()
    a.extra = [x for x in extra if x != "--"]
    _limit_pools(a.pure)
    import numpy as np
    import torch
    ref = os.path.abspath(a.gnn_dir)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    sys.path.insert(0, ref)
    os.chdir(ref)
    import stage_ref
    stage_ref.shim_reference_startup_bugs()
    if not a.pure:
        import gnnrag_amd  # noqa: F401
        from gnnrag_amd import install
        install.install()
        install.limit_host_threads()
    import parsing
    parser = argparse.ArgumentParser()
    parsing.add_parse_args(parser)
    argv = list(stage_ref.variant_argv(a.variant))
    argv[argv.index("--batch_size") + 1] = str(a.batch)
    import tempfile
    ck = tempfile.mkdtemp(prefix="gnnrag_train_") + "/"
    args = parser.parse_args(argv + ["--checkpoint_dir", ck, "--experiment_name", "timing"] + list(a.extra))
    args.use_cuda = (not a.pure) and torch.cuda.is_available()
    if not a.pure and not args.use_cuda:
        raise SystemExit("tools/time_train_step.py: the GPU leg needs a GPU (the package has no CPU path); --pure is the CPU leg")
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    from train_model import Trainer_KBQA
    from utils import create_logger
    tr = Trainer_KBQA(args=vars(args), model_name=args.model_name, logger=create_logger(args))
    tr.load_ckpt(os.path.join(stage_ref.CKPT, stage_ref.ckpt_name(a.variant)))
    if not a.pure:
        from gnnrag_amd.data.fact_mat import patch_loader
        patch_loader(tr.train_data, cache=False, keep_rng_stream=True)
    sync = (lambda: torch.cuda.synchronize()) if args.use_cuda else (lambda: None)
    # the body of Trainer_KBQA.train_epoch (train_model.py:209-233), statement for statement, with clocks at its seams
    tr.model.train()
    tr.train_data.reset_batches(is_sequential=False)
    names = ("get_batch", "forward", "backward", "clip_and_adam")
    steps, losses = [], []
    for it in range(a.warm + a.steps):
        sync()
        t = [time.perf_counter()]
        batch = tr.train_data.get_batch(it, tr.args["batch_size"], tr.args["fact_drop"])
        t.append(time.perf_counter())
        tr.optim_model.zero_grad()
        loss, _, _, tp_list = tr.model(batch, training=True)
        sync()
        t.append(time.perf_counter())
        loss.backward()
        sync()
        t.append(time.perf_counter())
        torch.nn.utils.clip_grad_norm_([p for _, p in tr.model.named_parameters()], tr.args["gradient_clip"])
        tr.optim_model.step()
        losses.append(loss.item())
        sync()
        t.append(time.perf_counter())
        steps.append([t[i + 1] - t[i] for i in range(4)])
    timed = np.array(steps[a.warm:]) * 1e3
    tot = timed.sum(1)
    native = None
    if not a.pure:
        native = [l.split()[-1] for l in open("/proc/self/maps") if "libgnnrag_hip" in l][:1]
    out = {"pure_reference": a.pure, "variant": a.variant, "batch_size": int(tr.args["batch_size"]), "steps": a.steps, "warm": a.warm,
           "ms_per_step": float(np.median(tot)), "ms_min": float(tot.min()), "ms_max": float(tot.max()),
           "stages_ms": {n: float(np.median(timed[:, i])) for i, n in enumerate(names)},
           "losses": [float(x) for x in losses], "threads": torch.get_num_threads(),
           "linear_dropout": float(tr.args["linear_dropout"]), "lm_dropout": float(tr.args["lm_dropout"]),
           "entity_dim": int(tr.args["entity_dim"]), "num_iter": int(tr.args["num_iter"]), "num_gnn": int(tr.args["num_gnn"]),
           "num_ins": int(tr.args["num_ins"]), "train_questions": int(tr.train_data.num_data),
           "padded_nodes_per_question": int(tr.train_data.max_local_entity), "native_library": native,
           "layer_class": type(tr.model.reasoning).__module__}
    import shutil
    shutil.rmtree(ck, ignore_errors=True)
    print("GNNRAG_TRAIN " + json.dumps(out))


if __name__ == "__main__":
    main()
