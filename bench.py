#!/usr/bin/env python
"""Benchmark of the GNN-RAG reasoning hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic question subgraphs:
L consecutive ReasonGNNLayer.forward calls (relation transform + typed-edge aggregation +
gated update + score + masked softmax each) starting from the seed distribution, exactly as
one iteration of ReaRev.forward drives the layer (reference rearev.py:206-211).  Inputs
(CSR structure, features, parameters) are resident in HBM when the timed region starts; the
CSR build is timed separately and reported as ``csr_build_ms``.

Workload (BASELINE.json configs[1], the config the metric is quoted on): synthetic
Freebase-shaped subgraphs, 2000 nodes / 10000 typed edges (+2000 self loops) per question,
batch 64 per GPU, 3 layers, hidden 200, 2 instructions, fp32.

With N > 1 every rank owns its own 64 questions (question-sharded, weak scaling) and ends
each step with ONE all-gather of the scored nodes (pred_dist) over RCCL.

Prints ONE JSON line (rank 0).  value = typed KG edges aggregated per second, whole job:
N * B * E * L * K / t  (t = max over ranks of the barrier-bracketed wall time of K steps).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable
FP32_MFMA_PEAK_TFLOPS = 157.3


def bytes_agg(cfg, F_g: int) -> float:
    """Algorithmic HBM bytes of ONE aggregation layer call (pinned):
    two CSRs (row_ptr + src + rel, int32), dist read, agg write, T tables once, instructions."""
    per_graph = 2 * ((cfg.N + 1) + 2 * F_g) * 4 + cfg.N * 4 + cfg.N * 2 * cfg.I * cfg.D * 4
    total = cfg.B * per_graph + 2 * cfg.R1 * cfg.D * 4 + cfg.B * cfg.I * cfg.D * 4
    if cfg.normalized_gnn:
        total += 2 * F_g * 4 * cfg.B
    return float(total)


def flops_update(cfg) -> float:
    return 2.0 * cfg.B * cfg.N * (2 * cfg.I + 1) * cfg.D * cfg.D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-b", type=int, default=8)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import _lib, ops, shard, stack, synth
    _lib.load()

    cfg = synth.CONFIGS[args.workload]
    # every rank owns its own questions (weak scaling): same shapes, different seed
    batch = synth.make_batch(cfg, seed=cfg.seed + 1000 * rank)
    feats = synth.make_features(cfg, seed=cfg.seed + 1000 * rank)
    params = synth.make_layer_params(cfg)
    F = batch.F
    F_g = F // cfg.B

    devin = stack.DeviceInputs(batch, feats, dev)
    layer = stack.build_layer(cfg, batch, params, dev)
    t0 = time.perf_counter()
    stack.init_reason(layer, batch, devin, devin.h0)          # uploads int32 tuple + device CSR build
    torch.cuda.synchronize()
    csr_first_ms = (time.perf_counter() - t0) * 1e3
    et = batch.edge_tuple
    t0 = time.perf_counter()
    for _ in range(3):
        ops.CsrPlan(et[0], et[1], et[2], cfg.B, cfg.N, cfg.R1, dev, validate=False)
    torch.cuda.synchronize()
    csr_build_ms = (time.perf_counter() - t0) * 1e3 / 3

    def step():
        layer.local_entity_emb = devin.h0
        d, _ = stack.run_layers(layer, cfg, devin)
        if distributed:
            d = shard.gather_rows(d, cfg.B * world)
        return d

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(last).all()

    ms_per_step = elapsed * 1e3 / args.steps
    typed_edges = cfg.B * cfg.E * cfg.L * world
    facts = F * cfg.L * world
    value = typed_edges / (elapsed / args.steps)

    out = {
        "metric": "kg_edges_aggregated_per_sec", "value": value, "unit": "typed-edge*layers/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s: synthetic Freebase-shaped subgraphs, %d nodes / %d typed edges (+%d self loops) "
                               "per question, batch %d per GPU, %d layers, hidden %d, %d instructions"
                               % (cfg.name, cfg.N, cfg.E, cfg.N, cfg.B, cfg.L, cfg.D, cfg.I),
                   "B_per_gpu": cfg.B, "N": cfg.N, "E": cfg.E, "F_per_question": F_g, "R1": cfg.R1,
                   "D": cfg.D, "I": cfg.I, "L": cfg.L, "parallelism": "question-sharded x%d" % world},
        "fact_layers_per_sec": facts / (elapsed / args.steps),
        "csr_build_ms": csr_build_ms, "csr_first_call_ms": csr_first_ms,
    }

    if rank == 0:
        out.update(roofline_leg(cfg, layer, devin, ops, F_g, args.steps))
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg(cfg, args.cpu_sample_b)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def _events_ms(fn, reps):
    """Average device time of fn() over reps launches, HIP events on torch's current stream
    (the stream the library launches on)."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


def roofline_leg(cfg, layer, devin, ops, F_g, steps):
    """Per-kernel device times with HIP events, op by op (same kernels, same arguments as the
    fused gnnrag_reason_layer call): the aggregation against the HBM roofline with the pinned
    algorithmic byte count, the dense update against the fp32 MFMA roofline."""
    B, N, D, I, L = cfg.B, cfg.N, cfg.D, cfg.I, cfg.L
    plan = layer.plan
    t_agg, t_agg_dense, t_upd, t_sm, t_rel = [], [], [], [], []
    reps = max(3, min(steps, 10))
    for _ in range(reps):
        h = devin.h0.reshape(B * N, D)
        dist = devin.seed_dist
        for j in range(L):
            rl = getattr(layer, "rel_linear%d" % j)
            e2e = getattr(layer, "e2e_linear%d" % j)
            box = {}

            def rel():
                box["Tf"] = ops.linear(devin.rel_features, rl.weight, rl.bias)
                box["Ti"] = ops.linear(devin.rel_features_inv, rl.weight, rl.bias)
            t_rel += _events_ms(rel, 1)

            def agg():
                box["agg"] = ops.aggregate(plan, dist, devin.ins[0], box["Tf"], box["Ti"])
            ms = _events_ms(agg, 1)
            t_agg += ms
            if j > 0:
                t_agg_dense += ms

            def upd():
                box["h"], box["score"] = ops.update_score(h, box["agg"], e2e.weight, e2e.bias,
                                                          layer.score_func.weight, layer.score_func.bias,
                                                          layer.local_entity_mask, I)
            t_upd += _events_ms(upd, 1)

            def sm():
                box["dist"] = ops.masked_softmax(box["score"], B, N)
            t_sm += _events_ms(sm, 1)
            h, dist = box["h"], box["dist"]

    # achievable streaming ceiling on this box (float4 copy: reads + writes)
    n = 256 * 1024 * 1024 // 4
    src = torch.empty(n, dtype=torch.float32, device=devin.h0.device).normal_()
    dst = torch.empty_like(src)
    ops.stream_copy(src, dst)
    tc = _events_ms(lambda: ops.stream_copy(src, dst), 10)
    copy_gbps = 2 * n * 4 / (np.median(tc) * 1e-3) / 1e9

    ba = bytes_agg(cfg, F_g)
    avg = float(np.mean(t_agg))
    avg_dense = float(np.mean(t_agg_dense))
    achieved = ba / (avg * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("aggregate_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    fl = flops_update(cfg)
    avg_upd = float(np.mean(t_upd))
    return {
        "roofline": {
            "kernel": "gnnrag_aggregate (k_walk_light + k_walk_heavy)", "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "algorithmic_bytes_per_launch": ba, "avg_launch_ms": avg,
            "launches_timed": len(t_agg),
            "note": "average over all layer calls of a step (first call of each iteration has the sparse seed "
                    "prior); dense-prior calls only: see dense_prior",
            "dense_prior": {"avg_launch_ms": avg_dense, "achieved": ba / (avg_dense * 1e-3) / 1e9,
                            "frac": ba / (avg_dense * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "measured_copy_ceiling_GBps": copy_gbps,
        },
        "roofline_update": {
            "kernel": "gnnrag_update_score (k_gemm_f32, fp32 MFMA 16x16x4)", "bound": "mfma",
            "achieved": fl / (avg_upd * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": fl / (avg_upd * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "avg_launch_ms": avg_upd,
            "flops_per_launch": fl,
        },
        "kernel_ms": {"rel_transform_x2": float(np.mean(t_rel)), "aggregate": avg, "update_score": avg_upd,
                      "softmax": float(np.mean(t_sm))},
    }


def cpu_baseline_leg(cfg, sample_b):
    """The reference's CPU op sequence (oracle/rearev_torch_cpu.py, a port: /root/reference is not
    on the GPU box) on a bounded sample of the same workload: sample_b questions of the same
    shape, all host cores, 1 warm-up + 2 timed L-layer passes."""
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import synth
    sub = synth.GraphConfig(**{**cfg.__dict__, "B": sample_b, "name": cfg.name + "-cpu-sample"})
    batch = synth.make_batch(sub)
    feats = synth.make_features(sub)
    params = synth.make_layer_params(sub)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    p = otorch.to_torch_params(params)
    relfeat = torch.from_numpy(feats["rel_features"])
    relfeat_inv = torch.from_numpy(feats["rel_features_inv"])
    ins = torch.from_numpy(feats["ins"][0])
    mask = torch.from_numpy((batch.local_entity != batch.num_entity).astype(np.float32))
    seed = torch.from_numpy(batch.seed_dist.astype(np.float32))
    t0 = time.perf_counter()
    st = otorch.Structure(batch.edge_tuple, sub.B, sub.N, sub.normalized_gnn)
    build_s = time.perf_counter() - t0

    def one_pass():
        h = torch.from_numpy(feats["h0"])
        dist = seed
        with torch.no_grad():
            for j in range(sub.L):
                _, dist, h = otorch.layer_forward(st, h, mask, dist, ins, p, j, relfeat, relfeat_inv, sub.pos_emb)
        return dist

    one_pass()
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        one_pass()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return {"value": sub.B * sub.E * sub.L / t, "unit": "typed-edge*layers/s", "cores": cores, "kind": "port",
            "sample": "%d questions of the same %s shape (N=%d, E=%d, D=%d, I=%d, L=%d), torch-CPU restatement of "
                      "the reference op sequence, 1 warm-up + 2 timed passes, median %.2f s/pass; "
                      "structure build %.2f s" % (sub.B, cfg.name, sub.N, sub.E, sub.D, sub.I, sub.L, t, build_s),
            "seconds_per_pass": t}


if __name__ == "__main__":
    main()
