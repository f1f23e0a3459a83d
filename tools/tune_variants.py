#!/usr/bin/env python
"""A/B harness for compile-time kernel variants: builds lib/exp_<name>.so per variant (CPU side,
`--build`), then (GPU side, `--run`) times the hot ops of workload C2 with each variant in its own
process and prints one line per variant.
    python tools/tune_variants.py --build          # here (hipcc cross-compiles)
    python tools/tune_variants.py --run            # on the GPU box (gpurun)
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# name -> extra -D switches ("__flags__": extra compiler flags, "__env__": environment of the run).  Variants that were
# measured and rejected are NOT kept as macros in the kernels: their numbers are in DESIGN.md (appendix) / profiles/.
VARIANTS = {
    "default": {},
    # macros that exist in the kernel sources today (add a macro to a kernel, list its values here, run):
    "gemm_nw4": {"GNNRAG_GEMM_MT1_NW": 4},                 # gemm_f32.hip: waves per workgroup of the k-tiled kernel
    "no_wres": {"GNNRAG_GEMM_WRES": 0},                    # the k-tiled kernel instead of the W-resident one (exact fp32)
    "no_split_tail": {"GNNRAG_SLICE_SPLIT_TAIL": 0},       # aggregate.hip, LDS walk
    "no_halfstep": {"GNNRAG_SLICE_HALFSTEP": 0},
    "wide_always": {"GNNRAG_SLICE_WIDE_LDS_KB": 159},
    "sl_unmerged": {"GNNRAG_SLICE_MERGED": 0},
    "sl_g2": {"GNNRAG_SLICE_BL_GROUP": 2}, "sl_g4": {"GNNRAG_SLICE_BL_GROUP": 4},
    # LDS walk workgroup sizes (two workgroups per CU either way): 512 / 640 / 768 threads = 4 / 5 / 6 waves per SIMD
    "sl_t512_g1": {"GNNRAG_SLICE_THREADS": 512, "GNNRAG_SLICE_WPE": 4, "GNNRAG_SLICE_BL_GROUP": 1},
    "sl_t768": {"GNNRAG_SLICE_THREADS": 768, "GNNRAG_SLICE_WPE": 6},
    "sl_t640": {"GNNRAG_SLICE_THREADS": 640, "GNNRAG_SLICE_WPE": 5},
    "vq_un2": {"GNNRAG_VQ_UN": 2}, "vq_un4": {"GNNRAG_VQ_UN": 4},      # tables_b3.hip: V pieces in flight while staging
    # gather walk (tables larger than LDS, BASELINE config 5; GNNRAG_TUNE_WORKLOAD=C5)
    "light_npw1": {"GNNRAG_LIGHT_NPW": 1}, "light_npw4": {"GNNRAG_LIGHT_NPW": 4},
    "quad_off": {"GNNRAG_LIGHT_QUAD": 0}, "quad_s2": {"GNNRAG_QUAD_STEPS": 2}, "quad_unmerged": {"GNNRAG_QUAD_MERGED": 0},
    "hub_u4": {"GNNRAG_HUB_U": 4}, "hub_ks16": {"GNNRAG_HUB_KS_MAX": 16}, "hub_ks4": {"GNNRAG_HUB_KS_MAX": 4},
    "hub_wg8192": {"GNNRAG_HUB_W_GRID": 8192}, "hub_wg512": {"GNNRAG_HUB_W_GRID": 512},
    "noslp": {"__flags__": ["-fno-slp-vectorize"]},        # the compiler SLP-packs the split's subtractions into v_pk_add_f32
    "split_trunc": {"GNNRAG_SPLIT_RN": 0},                 # the truncation form of the exact 3-way bf16 split (rounds 1-2)
    # the product library as built (lib/libgnnrag_hip.so) against lib/exp_default.so from an earlier build
    "mainlib": {"__env__": {"GNNRAG_LIB": os.path.join(REPO, "gnn-rag_amd", "lib", "libgnnrag_hip.so")}},
}

CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import gnnrag_amd
from gnnrag_amd import ops, stack, synth
import bench
dev = torch.device("cuda", 0)
cfg = synth.CONFIGS[os.environ.get("GNNRAG_TUNE_WORKLOAD", "C2")]
batch = synth.make_batch(cfg); feats = synth.make_features(cfg); params = synth.make_layer_params(cfg)
devin = stack.DeviceInputs(batch, feats, dev)
layer = stack.build_layer(cfg, batch, params, dev)
stack.init_reason(layer, batch, devin, devin.h0)
B, N, D, I = cfg.B, cfg.N, cfg.D, cfg.I
with torch.no_grad():
    dense, _ = layer(devin.seed_dist, devin.ins[0], step=0)
    rl, e2e, sf = layer.rel_linear1, layer.e2e_linear1, layer.score_func
    h = devin.h0.reshape(B * N, D)
    Tf = ops.linear(devin.rel_features, rl.weight, rl.bias); Ti = ops.linear(devin.rel_features_inv, rl.weight, rl.bias)
    P = ops.relation_tables(layer.plan, Tf, Ti, devin.ins[0], e2e.weight)
    nbr = ops.aggregate_fused(layer.plan, dense, P)
    agg = ops.aggregate(layer.plan, dense, devin.ins[0], Tf, Ti)
    ms = {}
    seed = devin.seed_dist
    only_upd = os.environ.get("GNNRAG_TUNE_ONLY") == "upd"      # the self-block update alone (short child)
    for name, prior in ((("aggf_dense", dense), ("aggf_seed", seed)) if not only_upd else ()):
        fn = lambda: ops.aggregate_fused(layer.plan, prior, P)
        fn()
        ms[name] = float(np.mean(bench._events_ms(fn, 20)))
    if os.environ.get("GNNRAG_TUNE_GEMM") and not only_upd:
        _, planes = ops.rel_transform(devin.rel_features, devin.rel_features_inv, [(rl.weight, rl.bias, None, None)], planes=True)
        fn = lambda: ops.relation_tables_planes(layer.plan, planes[0], devin.ins[0], e2e.weight)
        fn()
        ms["tables_vq"] = float(np.mean(bench._events_ms(fn, 10)))
    if only_upd:
        ops.set_dense_math(1)
        fn = lambda: ops.update_score_fused(h, nbr, e2e.weight, e2e.bias, sf.weight, sf.bias, layer.local_entity_mask, I)
        fn()
        for rep in range(3):
            ms["upd_fused_m1_%%d" %% rep] = float(np.mean(bench._events_ms(fn, 20)))
    for math in ((0, 1) if os.environ.get("GNNRAG_TUNE_GEMM") and not only_upd else ()):
        ops.set_dense_math(math)
        fns = {"tables": lambda: ops.relation_tables(layer.plan, Tf, Ti, devin.ins[0], e2e.weight),
               "upd": lambda: ops.update_score(h, agg, e2e.weight, e2e.bias, sf.weight, sf.bias, layer.local_entity_mask, I),
               "upd_fused": lambda: ops.update_score_fused(h, nbr, e2e.weight, e2e.bias, sf.weight, sf.bias, layer.local_entity_mask, I)}
        for name, fn in fns.items():
            fn()
            ms["%%s_m%%d" %% (name, math)] = float(np.mean(bench._events_ms(fn, 10)))
print("RESULT " + json.dumps(ms))
'''


def main():
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import build
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(VARIANTS)
    if "--build" in sys.argv:
        for n in names:
            if not VARIANTS[n] or set(VARIANTS[n]) - {"__env__"}:
                print(build.build_variant(n, VARIANTS[n]))
    if "--run" in sys.argv:
        for n in names:
            only_env = not (set(VARIANTS[n]) - {"__env__"})
            env = dict(os.environ, GNNRAG_LIB=os.path.join(build.LIBDIR, "exp_%s.so" % ("default" if only_env else n)))
            env.update(VARIANTS[n].get("__env__", {}))
            r = subprocess.run([sys.executable, "-c", CHILD % REPO], env=env, capture_output=True, text=True,
                               timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(n, "FAILED", r.stderr[-800:])
                continue
            ms = json.loads(line[0][7:])
            print("%-10s " % n + "  ".join("%s=%.1f" % (k[:18], v * 1e3) for k, v in ms.items()))


if __name__ == "__main__":
    main()
