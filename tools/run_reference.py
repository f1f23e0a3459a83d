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

    import parsing
    if not hasattr(parsing, "create_parser_nutrea"):
        parsing.create_parser_nutrea = lambda p: None

    # second reference bug: LSTMInstruction calls BaseInstruction.__init__(args) without the
    # `constraint` argument the base class requires; give it the default the other encoders pass
    from modules.question_encoding import base_encoder
    _orig_init = base_encoder.BaseInstruction.__init__

    def _init(self, args, constraint=False):
        _orig_init(self, args, constraint)
    base_encoder.BaseInstruction.__init__ = _init

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
                    patch_loader(dataset[split], cache=(split != "train"), keep_rng_stream=not is_eval, device=dev,
                                 structures=bool(os.environ.get("GNNRAG_DEVICE_STRUCTURES")) and dev is not None)
            return dataset

        dataset_load.load_data = load_data

    import evaluate
    if not os.environ.get("GNNRAG_NO_EVAL_PATCH"):
        from gnnrag_amd import eval_tail
        eval_tail.patch_evaluator_class(evaluate.Evaluator)
    if world > 1 or force_dist:
        from gnnrag_amd import shard
        _ev_init = evaluate.Evaluator.__init__

        def _init_sharded(self, *a, **kw):
            _ev_init(self, *a, **kw)
            shard.shard_model(self.model)                # evaluation batches are split over the ranks

        evaluate.Evaluator.__init__ = _init_sharded

    sys.argv = [os.path.join(ref, "main.py")] + sys.argv[2:]
    try:
        runpy.run_path(os.path.join(ref, "main.py"), run_name="__main__")
    finally:
        mapped = [l.split()[-1] for l in open("/proc/self/maps") if "libgnnrag_hip" in l]
        print("gnnrag_amd: native library %s" % ("mapped: " + mapped[0] if mapped else "NOT loaded"))


if __name__ == "__main__":
    main()
