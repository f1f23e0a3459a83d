#!/usr/bin/env python
"""Runs the UNMODIFIED reference entry point (GNN-RAG ``gnn/main.py``) with the MI355X path underneath.

    python tools/run_reference.py /path/to/GNN-RAG/gnn  ReaRev --is_eval --load_experiment X.ckpt ... (main.py's own flags)

What is substituted, all without touching a reference file:
  * ``modules.kg_reasoning.{reasongnn,base_gnn,nsm_gnn}``, ``modules.layer_init``, ``modules.query_update``
    -> this package's modules (``gnnrag_amd.install.install()``, before the reference imports its models);
  * every data loader the reference creates -> vectorised / cached ``_build_fact_mat``
    (``gnnrag_amd.data.fact_mat.patch_loader``; skip with GNNRAG_NO_LOADER_PATCH=1);
  * ``Evaluator.evaluate`` -> device-side candidate selection (``gnnrag_amd.eval_tail``; skip with
    GNNRAG_NO_EVAL_PATCH=1).
Under ``python -m torch.distributed.run --nproc-per-node G ... tools/run_reference.py ... --is_eval ...`` every
evaluation batch is question-sharded over the G GPUs (``gnnrag_amd.shard.shard_model``: local forward, one RCCL
all-gather of the scored nodes); rank 0 writes the ``.info`` file and prints the metrics.
``GNNRAG_E2E_TIMES=1`` prints one ``GNNRAG_E2E {json}`` line at exit: wall time and questions of every ``Evaluator.evaluate``
call and the time spent inside ``get_batch``, the structure build and ``Model.forward`` (bench.py's ``e2e`` block);
``GNNRAG_PURE_REFERENCE=1`` runs the reference WITHOUT any substitution (its own CPU / torch path: the baseline of that
block; start it with CUDA_VISIBLE_DEVICES="").
Needs a GPU (the package has no CPU path); the reference's own two start-up bugs (an undefined
``create_parser_nutrea`` in ``parsing.py``; ``LSTMInstruction`` not passing ``constraint``)
are shimmed exactly as the tests do."""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 2 or not os.path.isfile(os.path.join(sys.argv[1], "main.py")):
        raise SystemExit(__doc__)
    ref = os.path.abspath(sys.argv[1])
    sys.path.insert(0, REPO)
    sys.path.insert(0, ref)
    os.chdir(ref)
    pure = os.environ.get("GNNRAG_PURE_REFERENCE") == "1"
    if pure:
        return run_pure(ref)
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import install
    install.install()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    is_eval = "--is_eval" in sys.argv
    # GNNRAG_FORCE_DIST=1 on a single GPU: the question-sharded path (process group, shard_model, RCCL all-gather and
    # all-reduce) runs with world size 1 - what a 1-GPU box can prove of the multi-GPU path
    force_dist = os.environ.get("GNNRAG_FORCE_DIST") == "1" and world == 1 and is_eval
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("LOCAL_RANK", "0")
        os.environ["WORLD_SIZE"] = "1"
    if world > 1 or force_dist:                          # one process per GPU (torchrun), RCCL over xGMI
        if not is_eval:
            # only evaluation is question-sharded: a training run under torchrun would train one full replica per
            # rank without gradient sync, and every rank would write the same checkpoint / .info file
            raise SystemExit("tools/run_reference.py: with WORLD_SIZE > 1 only evaluation runs (--is_eval) are "
                             "supported; launch training as a single process")
        import atexit
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
        atexit.register(lambda: dist.is_initialized() and dist.destroy_process_group())

    # GNNRAG_MIOPEN_RNN=0: torch's cudnn / MIOpen backend off, i.e. torch's own per-step kernels for any nn.LSTM that is NOT
    # replaced by HipLSTM below (GNNRAG_HIP_LSTM=0, or a multi-layer / bidirectional encoder): MIOpen's RNN call takes ~12 ms
    # per call at the question encoder's shapes on the MI355X (tools/profile_e2e.sh).  A backend setting only.
    if os.environ.get("GNNRAG_MIOPEN_RNN") == "0":
        import torch
        torch.backends.cudnn.enabled = False
    # the reference's two start-up bugs: an undefined create_parser_nutrea; LSTMInstruction calls
    # BaseInstruction.__init__(args) without the `constraint` argument the base class requires
    shim_startup_bugs()

    if not os.environ.get("GNNRAG_NO_LOADER_PATCH"):
        import dataset_load
        from gnnrag_amd.data.fact_mat import patch_loader
        orig_load = dataset_load.load_data

        def load_data(*a, **kw):
            dataset = orig_load(*a, **kw)
            for split in ("train", "valid", "test"):
                if dataset.get(split) is not None:
                    # the per-question cache skips numpy RNG draws the reference makes (fact_mat.patch_loader):
                    # evaluation-only runs use it as it is, training runs keep the reference's RNG stream
                    # GNNRAG_DEVICE_FACTS=1 (single-rank evaluation): per-question id blocks stay on the GPU
                    dev = None
                    if ((os.environ.get("GNNRAG_DEVICE_FACTS") or os.environ.get("GNNRAG_DEVICE_STRUCTURES"))
                            and is_eval and split != "train"):
                        import torch
                        dev = torch.device("cuda", torch.cuda.current_device())
                    # GNNRAG_DEVICE_STRUCTURES=1: additionally every question's sorted structure stays on the GPU and a
                    # batch's structure is their concatenation (no per-batch sort)
                    # more than one rank: a rank builds the tuple of ITS questions only (fact_mat.ShardedFacts)
                    rk = (int(os.environ.get("RANK", "0")), world) if (world > 1 and is_eval and split != "train") else None
                    patch_loader(dataset[split], cache=(split != "train"), keep_rng_stream=not is_eval, device=dev,
                                 structures=bool(os.environ.get("GNNRAG_DEVICE_STRUCTURES")) and dev is not None,
                                 shard=rk)
            return dataset

        dataset_load.load_data = load_data

    import evaluate
    if not os.environ.get("GNNRAG_NO_EVAL_PATCH"):
        from gnnrag_amd import eval_tail
        eval_tail.patch_evaluator_class(evaluate.Evaluator)
    if os.environ.get("GNNRAG_HIP_LSTM", "1") != "0":
        # the question encoder's nn.LSTM -> HipLSTM (same parameters, shared; inference calls only): the model is built
        # by the reference's own code, the Evaluator's constructor is the first place that sees it
        _ev_init0 = evaluate.Evaluator.__init__

        def _init_lstm(self, *a, **kw):
            _ev_init0(self, *a, **kw)
            install.swap_lstm(self.model)

        evaluate.Evaluator.__init__ = _init_lstm
    if world > 1 or force_dist:
        from gnnrag_amd import shard
        _ev_init = evaluate.Evaluator.__init__

        def _init_sharded(self, *a, **kw):
            _ev_init(self, *a, **kw)
            shard.shard_model(self.model)                # evaluation batches are split over the ranks

        evaluate.Evaluator.__init__ = _init_sharded

    times = install_e2e_timers(pure=False) if os.environ.get("GNNRAG_E2E_TIMES") == "1" else None
    sys.argv = [os.path.join(ref, "main.py")] + sys.argv[2:]
    try:
        runpy.run_path(os.path.join(ref, "main.py"), run_name="__main__")
    finally:
        mapped = [l.split()[-1] for l in open("/proc/self/maps") if "libgnnrag_hip" in l]
        print("gnnrag_amd: native library %s" % ("mapped: " + mapped[0] if mapped else "NOT loaded"))
        if times is not None:
            import json
            if not os.environ.get("GNNRAG_NO_EVAL_PATCH"):
                times["retrieved"] = dict(eval_tail.STATS)      # candidates the Evaluator's own loop walked
            print("GNNRAG_E2E " + json.dumps(times))


def shim_startup_bugs():
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


def run_pure(ref):
    """The reference as it is (plus its two start-up shims), timed: bench.py's CPU baseline of the e2e block."""
    import json
    shim_startup_bugs()
    times = install_e2e_timers(pure=True) if os.environ.get("GNNRAG_E2E_TIMES") == "1" else None
    sys.argv = [os.path.join(ref, "main.py")] + sys.argv[2:]
    try:
        runpy.run_path(os.path.join(ref, "main.py"), run_name="__main__")
    finally:
        if times is not None:
            print("GNNRAG_E2E " + json.dumps(times))


def install_e2e_timers(pure: bool) -> dict:
    """Wall-clock timers around the stages of an evaluation run, at the reference's own seams: ``Evaluator.evaluate``
    (evaluate.py:147), ``get_batch`` (dataset_load.py:599), the structure build (``build_matrix`` of the reference,
    base_gnn.py:19 / this package's ``plan_for``) and ``Model.forward`` (rearev.py:163).  GPU stages are bracketed by
    device synchronisation, so the split is honest and the run a little slower than an untimed one."""
    import time
    import torch
    import dataset_load
    import evaluate
    T = {"evaluate_calls": [], "get_batch_s": 0.0, "forward_s": 0.0, "structure_s": 0.0, "batches": 0,
         "threads": torch.get_num_threads(), "pure_reference": pure}
    sync = (lambda: torch.cuda.synchronize()) if (torch.cuda.is_available() and not pure) else (lambda: None)

    def timed(fn, key, count=None, do_sync=False):
        def wrapper(*a, **kw):
            if do_sync:
                sync()
            t0 = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                if do_sync:
                    sync()
                T[key] += time.perf_counter() - t0
                if count:
                    T[count] += 1
        return wrapper

    dataset_load.SingleDataLoader.get_batch = timed(dataset_load.SingleDataLoader.get_batch, "get_batch_s")
    from models.ReaRev import rearev
    rearev.ReaRev.forward = timed(rearev.ReaRev.forward, "forward_s", "batches", do_sync=True)
    if pure:
        from modules.kg_reasoning import base_gnn
        base_gnn.BaseGNNLayer.build_matrix = timed(base_gnn.BaseGNNLayer.build_matrix, "structure_s")
    else:
        from gnnrag_amd.modules.kg_reasoning import base_gnn as g_base
        import gnnrag_amd.modules.layer_init as g_li
        g_base.plan_for = g_li.plan_for = timed(g_base.plan_for, "structure_s", do_sync=True)   # TypeLayer builds it first
    ev = evaluate.Evaluator.evaluate

    def evaluate_timed(self, valid_data, *a, **kw):
        before = (T["get_batch_s"], T["forward_s"], T["structure_s"], T["batches"])
        sync()
        t0 = time.perf_counter()
        out = ev(self, valid_data, *a, **kw)
        sync()
        dt = time.perf_counter() - t0
        T["evaluate_calls"].append({"questions": int(valid_data.num_data), "seconds": dt,
                                    "get_batch_s": T["get_batch_s"] - before[0], "forward_s": T["forward_s"] - before[1],
                                    "structure_s": T["structure_s"] - before[2], "batches": T["batches"] - before[3],
                                    "max_local_entity": int(valid_data.max_local_entity)})
        return out
    evaluate.Evaluator.evaluate = evaluate_timed
    return T


if __name__ == "__main__":
    main()
