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


def _limit_thread_pools_early():
    """Before numpy / torch are imported (same rule as bench.py): OpenMP / OpenBLAS / MKL pools of min(8, CPU quota / 2)
    threads unless the variables are already set - pools sized after the visible hardware threads spin a container's CPU
    quota away and get every thread of the process parked (gnnrag_amd.install.limit_host_threads).  Not for the pure
    reference run (its thread count is the caller's: bench.py sets OMP_NUM_THREADS for that leg) and not with
    GNNRAG_HOST_THREADS=0."""
    if os.environ.get("GNNRAG_HOST_THREADS") == "0" or os.environ.get("GNNRAG_PURE_REFERENCE") == "1":
        return
    budget = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            budget = min(budget, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    n = os.environ.get("GNNRAG_HOST_THREADS") or str(max(1, min(8, budget // 2)))
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(k, n)


_limit_thread_pools_early()


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
    # the process that drives the GPU needs no OpenMP team (install.limit_host_threads: spinning workers exhaust a
    # container's CPU quota and the kernel parks the launching thread with them); GNNRAG_HOST_THREADS=0 leaves torch's
    # default, any other number sets that many
    if os.environ.get("GNNRAG_HOST_THREADS", "") != "0":
        install.limit_host_threads(int(os.environ["GNNRAG_HOST_THREADS"]) if os.environ.get("GNNRAG_HOST_THREADS") else None)

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
            # the reference's per-candidate evaluation tail in a process of its own (gnnrag_amd.eval_tail.start_tail_server):
            # forked HERE - the splits are loaded, the GPU has not been touched yet (a fork behind a live ROCm context slows
            # every host-to-device copy of the parent); GNNRAG_EVAL_PIPELINE=0 keeps the tail in this process
            if is_eval and world == 1 and not force_dist and not os.environ.get("GNNRAG_NO_EVAL_PATCH"):
                from gnnrag_amd import eval_tail as _et
                _et.start_tail_server(dataset)
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

    freeze_after_setup(evaluate)
    times = install_e2e_timers(pure=False) if os.environ.get("GNNRAG_E2E_TIMES") == "1" else None
    if os.environ.get("GNNRAG_PROFILE_FORWARD"):
        install_forward_profile(os.path.abspath(os.path.join(REPO, os.environ["GNNRAG_PROFILE_FORWARD"])),
                                os.environ.get("GNNRAG_PROFILE_KIND", "torch"))
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


def freeze_after_setup(evaluate) -> None:
    """``gc.freeze()`` once the data is loaded and the model built (``Evaluator.__init__`` is the last step of
    ``Trainer_KBQA.__init__``, train_model.py:25-72): see ``gnnrag_amd.install.freeze_loaded_data``.  Applied to the pure
    reference run as well (same interpreter-level setting for both sides of the e2e comparison); GNNRAG_GC_FREEZE=0
    leaves the collector as it is."""
    if os.environ.get("GNNRAG_GC_FREEZE", "1") == "0":
        return
    init = evaluate.Evaluator.__init__

    def _init(self, *a, **kw):
        init(self, *a, **kw)
        import gc
        gc.collect()
        gc.freeze()
    evaluate.Evaluator.__init__ = _init


def run_pure(ref):
    """The reference as it is (plus its two start-up shims), timed: bench.py's CPU baseline of the e2e block."""
    import json
    shim_startup_bugs()
    import evaluate
    freeze_after_setup(evaluate)
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


def install_forward_profile(out_path: str, kind: str, warm: int = None, calls: int = None) -> None:
    """GNNRAG_PROFILE_FORWARD=<file> [GNNRAG_PROFILE_KIND=torch|cprofile]: where the HOST time of a steady-state
    ``ReaRev.forward`` goes.  Calls ``warm`` .. ``warm + calls - 1`` run under torch.profiler (CPU side of every op and
    launch) or cProfile (Python functions); first-call costs (kernel loading, hipBLASLt heuristics, allocator growth) stay
    outside.  The summary is written when the last profiled call returns."""
    import io
    import time
    import torch
    from models.ReaRev import rearev
    warm = int(os.environ.get("GNNRAG_PROFILE_WARM", "8")) if warm is None else warm
    calls = int(os.environ.get("GNNRAG_PROFILE_CALLS", "12")) if calls is None else calls
    fwd = rearev.ReaRev.forward
    state = {"n": 0, "prof": None, "wall": 0.0, "wall_sync": 0.0}
    if kind.startswith("sections"):
        # perf_counter around the forward's own seams, no profiler: host time per section ("sections"), or with a device
        # synchronisation at every seam ("sections_sync": the section's device work included)
        import gc
        from modules.question_encoding import base_encoder
        import gnnrag_amd.modules.kg_reasoning.reasongnn as g_rg
        import gnnrag_amd.modules.query_update as g_qu
        import gnnrag_amd.modules.layer_init as g_li
        sec, on, marks = {}, {"v": False}, {}
        do_sync = kind == "sections_sync"

        def wrap(cls, name, label):
            f = getattr(cls, name)

            def w(*a, **kw):
                if not on["v"]:
                    return f(*a, **kw)
                if do_sync:
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                marks.setdefault("first_in", t0)
                r = f(*a, **kw)
                if do_sync:
                    torch.cuda.synchronize()
                t1 = time.perf_counter()
                if not label.startswith("  "):
                    marks["last_out"] = t1
                d = sec.setdefault(label, [0.0, 0])
                d[0] += t1 - t0
                d[1] += 1
                return r
            setattr(cls, name, w)
        wrap(base_encoder.BaseInstruction, "init_reason", "instruction.init_reason (question encoder)")
        wrap(base_encoder.BaseInstruction, "get_instruction", "instruction.get_instruction")
        wrap(rearev.ReaRev, "init_reason", "ReaRev.init_reason (relation features, TypeLayer, structure)")
        wrap(g_rg.ReasonGNNLayer, "forward", "ReasonGNNLayer.forward")
        wrap(g_qu.QueryReform, "forward", "QueryReform.forward")
        wrap(g_li.TypeLayer, "forward", "  of which TypeLayer.forward")
        wrap(rearev.ReaRev, "calc_loss_label", "calc_loss_label")
        wrap(rearev.ReaRev, "get_rel_feature", "  of which get_rel_feature")
        wrap(rearev.ReaRev, "get_ent_init", "  of which get_ent_init")
        if os.environ.get("GNNRAG_PROFILE_GC") == "0":
            gc.disable()
        gc_log, gc_t0, per_call, allocs = [], {}, [], []

        def gc_cb(phase, info):          # every collection while a profiled forward runs: generation, duration, objects freed
            if phase == "start":
                gc_t0["t"] = time.perf_counter()
            elif on["v"]:
                gc_log.append((info["generation"], (time.perf_counter() - gc_t0.get("t", time.perf_counter())) * 1e3,
                               info["collected"], info["uncollectable"]))
        gc.callbacks.append(gc_cb)

        def forward(self, *a, **kw):
            n = state["n"]
            state["n"] += 1
            on["v"] = warm <= n < warm + calls
            if on["v"]:
                torch.cuda.synchronize()
            marks.clear()
            ms0 = torch.cuda.memory_stats() if on["v"] else None
            t0 = time.perf_counter()
            out = fwd(self, *a, **kw)
            if on["v"]:
                t1 = time.perf_counter()
                ms1 = torch.cuda.memory_stats()
                allocs.append((ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0),
                               ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0),
                               round(ms1.get("reserved_bytes.all.current", 0) / 2**20)))
                state["wall"] += t1 - t0
                torch.cuda.synchronize()
                state["wall_sync"] += time.perf_counter() - t0
                for label, dt in (("forward entry -> first module call (input tensors to the device)", marks["first_in"] - t0),
                                  ("last module call -> return (argmax of the last distribution)", t1 - marks["last_out"])):
                    d = sec.setdefault(label, [0.0, 0])
                    d[0] += dt
                    d[1] += 1
                per_call.append(round((t1 - t0) * 1e3, 2))
            on["v"] = False
            if n == warm + calls - 1:
                with open(out_path, "w") as f:
                    f.write("# %d steady-state ReaRev.forward calls (after %d warm ones), %s, gc %s: %.3f ms per call until "
                            "forward returns, %.3f ms until the device is idle\n" % (
                                calls, warm, kind, "on" if gc.isenabled() else "off", state["wall"] / calls * 1e3,
                                state["wall_sync"] / calls * 1e3))
                    for k, (t, c) in sec.items():
                        f.write("%-70s %8.3f ms per forward  (%d calls per forward)\n" % (k, t / calls * 1e3, c // calls))
                    f.write("per call, ms: %s\n" % per_call)
                    f.write("per call, caching allocator (device allocations, device frees, MiB reserved after): %s\n" % allocs)
                    f.write("collections inside the %d forwards: %d, %.2f ms in total; frozen objects %d; tracked objects now %d; "
                            "the long ones (generation, ms, collected, uncollectable): %s\n" % (
                                calls, len(gc_log), sum(x[1] for x in gc_log), gc.get_freeze_count(), len(gc.get_objects()),
                                [(g, round(ms, 2), c, u) for g, ms, c, u in gc_log if ms > 0.5]))
            return out
        rearev.ReaRev.forward = forward
        return

    def forward(self, *a, **kw):
        n = state["n"]
        state["n"] += 1
        if n == warm:
            torch.cuda.synchronize()
            if kind == "cprofile":
                import cProfile
                state["prof"] = cProfile.Profile()
                state["prof"].enable()
            else:
                acts = [torch.profiler.ProfilerActivity.CPU]
                if kind == "torchcuda":
                    acts.append(torch.profiler.ProfilerActivity.CUDA)
                state["prof"] = torch.profiler.profile(activities=acts)
                state["prof"].__enter__()
        t0 = time.perf_counter()
        out = fwd(self, *a, **kw)
        if warm <= n < warm + calls:
            state["wall"] += time.perf_counter() - t0
            torch.cuda.synchronize()
            state["wall_sync"] += time.perf_counter() - t0
        if n == warm + calls - 1:
            prof = state["prof"]
            buf = io.StringIO()
            buf.write("# %d steady-state ReaRev.forward calls (after %d warm ones), host wall %.3f ms per call until forward "
                      "returns, %.3f ms until the device is idle\n" % (calls, warm, state["wall"] / calls * 1e3,
                                                                        state["wall_sync"] / calls * 1e3))
            if kind == "cprofile":
                import pstats
                prof.disable()
                for key in ("cumulative", "tottime"):
                    st = pstats.Stats(prof, stream=buf)
                    st.sort_stats(key).print_stats(45)
            else:
                prof.__exit__(None, None, None)
                buf.write(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=50, max_name_column_width=70))
                if kind == "torchcuda":
                    buf.write("\n\n==== by device time ====\n")
                    buf.write(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=90))
            torch.cuda.synchronize()
            with open(out_path, "w") as f:
                f.write(buf.getvalue())
        return out
    rearev.ReaRev.forward = forward


if __name__ == "__main__":
    main()
