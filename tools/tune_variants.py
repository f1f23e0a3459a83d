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

# name -> extra -D switches.  Switches that exist today: GNNRAG_REASON_SLICE, GNNRAG_SLICE_SPLIT_TAIL,
# GNNRAG_SLICE_HALFSTEP (aggregate.hip), GNNRAG_GEMM_MT1_NW (gemm_f32.hip).  Add a macro to the kernel source, list its values here, run.
VARIANTS = {
    "default": {},
    "reason_slice": {"GNNRAG_REASON_SLICE": 1},
    "gemm_nw4": {"GNNRAG_GEMM_MT1_NW": 4},
    "no_split_tail": {"GNNRAG_SLICE_SPLIT_TAIL": 0},
    "no_halfstep": {"GNNRAG_SLICE_HALFSTEP": 0},
    "wide_always": {"GNNRAG_SLICE_WIDE_LDS_KB": 159},
    # timing-only ablations of k_gemm_f32 (results are wrong on purpose): see GNNRAG_GEMM_ABL in gemm_f32.hip
    "abl_nomfma": {"GNNRAG_GEMM_ABL": 1},
    "abl_noepi": {"GNNRAG_GEMM_ABL": 2},
    "abl_noload": {"GNNRAG_GEMM_ABL": 4},
    "abl_nolds": {"GNNRAG_GEMM_ABL": 8},
    "abl_mfma_epi": {"GNNRAG_GEMM_ABL": 12},
    "abl_mfma_only": {"GNNRAG_GEMM_ABL": 14},
    "abl_mem_only": {"GNNRAG_GEMM_ABL": 1 + 8},
    "timing": {"GNNRAG_GEMM_TIMING": 1},        # per-wave phase stamps of k_gemm_wres (tools/gemm_timeline_wres.py)
    "no_wres": {"GNNRAG_GEMM_WRES": 0},         # the k-tiled kernel for the self-block update
    # timing-only ablations of k_tables_vq / k_update_b3 (tables_b3.hip).  CAUTION: the *_nolds variants feed the A
    # fragments in place of the LDS weight fragments, which makes the column tiles' MFMA chains identical - the compiler
    # merges them (4x fewer MFMAs): they bound nothing.  upd_halfmfma (3 of 6 plane products) is the honest MFMA probe.
    # k_walk_slice (aggregate.hip): 1 no table staging loads, 2 no output stores, 8 no pair loads; GNNRAG_TUNE_WORKLOAD=C2u
    # runs the same on uniformly drawn heads (what the degree skew costs)
    "sl_unmerged": {"GNNRAG_SLICE_MERGED": 0},
    "sl_branchy": {"GNNRAG_SLICE_BRANCHLESS": 0}, "sl_g1": {"GNNRAG_SLICE_BL_GROUP": 1}, "sl_g2": {"GNNRAG_SLICE_BL_GROUP": 2},
    "sl_g4": {"GNNRAG_SLICE_BL_GROUP": 4},
    "sl_nostage": {"GNNRAG_SLICE_ABL": 1}, "sl_nostore": {"GNNRAG_SLICE_ABL": 2},
    "sl_nopairs": {"GNNRAG_SLICE_ABL": 8}, "sl_nomem": {"GNNRAG_SLICE_ABL": 11},
    # LDS walk with 512-thread workgroups (still two per CU: 4 waves per SIMD, 128 VGPRs) and 1 / 2 / 4 table rows in flight
    "sl_t512_g1": {"GNNRAG_SLICE_THREADS": 512, "GNNRAG_SLICE_WPE": 4, "GNNRAG_SLICE_BL_GROUP": 1},
    "sl_t512_g2": {"GNNRAG_SLICE_THREADS": 512, "GNNRAG_SLICE_WPE": 4, "GNNRAG_SLICE_BL_GROUP": 2},
    "sl_t512_g4": {"GNNRAG_SLICE_THREADS": 512, "GNNRAG_SLICE_WPE": 4, "GNNRAG_SLICE_BL_GROUP": 4},
    "sl_t1024_w4_g4": {"GNNRAG_SLICE_WPE": 4, "GNNRAG_SLICE_BL_GROUP": 4},
    "vq_un4": {"GNNRAG_VQ_UN": 4}, "vq_un5": {"GNNRAG_VQ_UN": 5}, "vq_un6": {"GNNRAG_VQ_UN": 6},
    # gather walk (tables larger than LDS, BASELINE config 5; GNNRAG_TUNE_WORKLOAD=C5): nodes per lane group whose
    # structure loads are requested together, workgroups of the hub-row kernel
    "light_npw1": {"GNNRAG_LIGHT_NPW": 1}, "light_npw2": {"GNNRAG_LIGHT_NPW": 2}, "light_npw8": {"GNNRAG_LIGHT_NPW": 8},
    "heavy_grid512": {"GNNRAG_HEAVY_GRID": 512}, "heavy_grid2048": {"GNNRAG_HEAVY_GRID": 2048},
    "light_nogather": {"GNNRAG_LIGHT_ABL": 1}, "light_nostore": {"GNNRAG_LIGHT_ABL": 2}, "light_l2hit": {"GNNRAG_LIGHT_ABL": 4},
    "quad_off": {"GNNRAG_LIGHT_QUAD": 0}, "quad_s2": {"GNNRAG_QUAD_STEPS": 2}, "quad_s3": {"GNNRAG_QUAD_STEPS": 3}, "quad_s4": {"GNNRAG_QUAD_STEPS": 4},
    "quad_npw1": {"GNNRAG_LIGHT_NPW": 1}, "quad_npw4": {"GNNRAG_LIGHT_NPW": 4},
    "quad_unmerged": {"GNNRAG_QUAD_MERGED": 0},
    "hub_u4": {"GNNRAG_HUB_U": 4}, "hub_u3": {"GNNRAG_HUB_U": 3}, "hub_ks16": {"GNNRAG_HUB_KS_MAX": 16},
    "hub_ks16_u4": {"GNNRAG_HUB_KS_MAX": 16, "GNNRAG_HUB_U": 4}, "hub_ks4_u4": {"GNNRAG_HUB_KS_MAX": 4, "GNNRAG_HUB_U": 4},
    "light_nomem": {"GNNRAG_LIGHT_ABL": 3}, "hub_ks32": {"GNNRAG_HUB_KS_MAX": 32}, "hub_ks8": {"GNNRAG_HUB_KS_MAX": 8}, "hub_ks4": {"GNNRAG_HUB_KS_MAX": 4}, "hub_wg8192": {"GNNRAG_HUB_W_GRID": 8192}, "hub_wg512": {"GNNRAG_HUB_W_GRID": 512},
    # round 4: the compiler SLP-packs the split's subtractions into v_pk_add_f32, which is expensive beside MFMAs
    "noslp": {"__flags__": ["-fno-slp-vectorize"]},
    # round 4: the self-block update with 12 / 16 waves per workgroup (k_update_b3w)
    "updw12": {"GNNRAG_UPD_WAVES": 12}, "updw16": {"GNNRAG_UPD_WAVES": 16},
    # the register-resident update (update_wr.hip; measured slower, opt-in) on at run time ("__env__": the default build
    # with these environment variables)
    "wr_on": {"__env__": {"GNNRAG_UPDATE_WR": "1"}},
    # k_update_b3 with the second wave of every SIMD started ~1/4, 1/2, 1 tile late (phase-locking experiment)
    "upd_balanced": {"GNNRAG_UPD_BALANCE": 1},         # k_update_b3 with 7 : 6 row chunks for its two column parts (slower)
    "vq_order1": {"GNNRAG_VQ_ORDER": 1},     # k_tables_vq: V fragments shared by the wave's row tiles
    "prio2": {"GNNRAG_UPD_PRIO": 2}, "prio2_desync80": {"GNNRAG_UPD_PRIO": 2, "GNNRAG_UPD_DESYNC": 80},
    "desync40": {"GNNRAG_UPD_DESYNC": 40}, "desync80": {"GNNRAG_UPD_DESYNC": 80}, "desync160": {"GNNRAG_UPD_DESYNC": 160},
    "split_trunc": {"GNNRAG_SPLIT_RN": 0},      # the truncation form of the exact 3-way bf16 split (rounds 1-2)
    # round 5: the LDS walk with three named pipeline stages (no stage copies) at 1024 / 768 / 640 threads per workgroup
    # (two workgroups per CU: 8 / 6 / 5 waves per SIMD, 64 / 80 / 96 registers)
    "sl_named": {"GNNRAG_SLICE_NAMED": 1},
    "sl_named_t768": {"GNNRAG_SLICE_NAMED": 1, "GNNRAG_SLICE_THREADS": 768, "GNNRAG_SLICE_WPE": 6},
    "sl_named_t640": {"GNNRAG_SLICE_NAMED": 1, "GNNRAG_SLICE_THREADS": 640, "GNNRAG_SLICE_WPE": 5},
    "sl_named_t640_g2": {"GNNRAG_SLICE_NAMED": 1, "GNNRAG_SLICE_THREADS": 640, "GNNRAG_SLICE_WPE": 5, "GNNRAG_SLICE_BL_GROUP": 2},
    "sl_t768": {"GNNRAG_SLICE_THREADS": 768, "GNNRAG_SLICE_WPE": 6},
    "sl_t640": {"GNNRAG_SLICE_THREADS": 640, "GNNRAG_SLICE_WPE": 5},
    # round 5: the self-block update on 32x32x16 MFMAs (update_x32.hip) switched off at run time
    "x32_off": {"__env__": {"GNNRAG_UPDATE_X32": "0"}},
    "x32_f1": {"__env__": {"GNNRAG_UPDATE_X32": "1"}},       # form 1: two waves per SIMD
    "x32_f2": {"__env__": {"GNNRAG_UPDATE_X32": "2"}},       # form 2: one wave per SIMD, source-pipelined, pinned interleave
    "x1_valu2": {"GNNRAG_X1_VALU": 2, "GNNRAG_X32_DEFAULT": 2}, "x1_valu4": {"GNNRAG_X1_VALU": 4, "GNNRAG_X32_DEFAULT": 2},
    "x1_valu6": {"GNNRAG_X1_VALU": 6, "GNNRAG_X32_DEFAULT": 2},
    "vq_nolds": {"GNNRAG_VQ_ABL": 1}, "vq_noa": {"GNNRAG_VQ_ABL": 2}, "vq_nostage": {"GNNRAG_VQ_ABL": 4},
    "upd_nolds": {"GNNRAG_UPD_ABL": 1}, "upd_noa": {"GNNRAG_UPD_ABL": 2}, "upd_noadd": {"GNNRAG_UPD_ABL": 4},
    "upd_nostore": {"GNNRAG_UPD_ABL": 8}, "upd_nosplit": {"GNNRAG_UPD_ABL": 16}, "upd_mfma_only": {"GNNRAG_UPD_ABL": 31},
    "upd_nomem": {"GNNRAG_UPD_ABL": 14}, "upd_halfmfma": {"GNNRAG_UPD_ABL": 32}, "upd_halfmfma_nomem": {"GNNRAG_UPD_ABL": 32 + 14}, 
    "vq_noepi": {"GNNRAG_VQ_ABL": 8}, "vq_mfma_only": {"GNNRAG_VQ_ABL": 15}, "vq_mfma_epi": {"GNNRAG_VQ_ABL": 7}, "vq_un2": {"GNNRAG_VQ_UN": 2},
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
            env = dict(os.environ, GNNRAG_LIB=os.path.join(build.LIBDIR, "exp_%s.so" % ("default" if only_env else n)),
                       **VARIANTS[n].get("__env__", {}))
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
