#!/usr/bin/env python
"""profiles/pmc_traffic.json from rocprofv3 PMC databases (FETCH_SIZE / WRITE_SIZE passes of
tools/prof_ops.py --ops agg,aggf).  HBM bytes per aggregation launch = sum over the kernels of one
launch of WRITE_SIZE + FETCH_SIZE (KB units -> bytes).  FETCH_SIZE on gfx950 under-reports wide
coalesced streams by 2x (MI355X_MICROARCH.md, HBM section): both the raw and the doubled figure
are stored, `*_hbm_bytes_per_launch` uses the doubled (conservative) one."""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    q = """select s.kernel_name, d.id, sum(e.value) from rocpd_pmc_event e
           join rocpd_info_pmc i on e.pmc_id = i.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id where i.name = ? group by d.id"""
    acc = defaultdict(list)
    for name, _, val in c.execute(q, (counter,)):
        acc[name].append(val)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def recompute(path):
    """Group totals of a stored result recomputed from its per-kernel tables (after a kernel was renamed or added)."""
    old = json.load(open(path))
    res = totals(old["per_kernel_fetch_KB"], old["per_kernel_write_KB"])
    for k in ("source", "workload", "commit", "kernel_sources_sha256", "fetch_correction"):
        res[k] = old[k]
    res["per_kernel_fetch_KB"], res["per_kernel_write_KB"] = old["per_kernel_fetch_KB"], old["per_kernel_write_KB"]
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k.endswith("per_launch")}))


def totals(fetch, write):
    res = {}

    def group(pred, launches_per_call=1.0):
        f = sum(v for k, v in fetch.items() if pred(k)) * 1024.0
        w = sum(v for k, v in write.items() if pred(k)) * 1024.0
        return f, w

    f, w = group(lambda k: "k_walk_slice" in k or "k_fact_prior" in k or ("k_walk_light" in k and "ILi2E" in k)
                 or "k_walk_light_q" in k or ("k_heavy" in k and "ILi2E" in k) or "k_hub_" in k)
    res.update(aggregate_fused_fetch_bytes_raw=f, aggregate_fused_write_bytes=w,
               aggregate_fused_hbm_bytes_per_launch=2 * f + w)
    # the seed-prior (frontier) form of layer 0: frontier + table rows + neighbour sums of the frontier
    f, w = group(lambda k: "k_frontier_build" in k or "k_tables_frontier" in k or "k_walk_frontier" in k)
    res.update(frontier_fetch_bytes_raw=f, frontier_write_bytes=w, frontier_hbm_bytes_per_launch=2 * f + w)
    f, w = group(lambda k: ("k_walk_light" in k or "k_heavy" in k) and "ILi0E" in k)
    res.update(aggregate_fetch_bytes_raw=f, aggregate_write_bytes=w, aggregate_hbm_bytes_per_launch=2 * f + w)
    # the self-block update with a dense nbr (k_update_b3<false>); its row-gated form of layer 0 (k_update_b3<true>, rows off
    # the frontier read a zero row) is a different launch and is kept apart
    f, w = group(lambda k: ("k_update_b3ILb0E" in k))
    if f or w:
        res.update(update_score_fused_fetch_bytes_raw=f, update_score_fused_write_bytes=w,
                   update_score_fused_hbm_bytes_per_launch=2 * f + w)
    f, w = group(lambda k: ("k_update_b3ILb1E" in k))
    if f or w:
        res.update(update_score_gated_fetch_bytes_raw=f, update_score_gated_write_bytes=w,
                   update_score_gated_hbm_bytes_per_launch=2 * f + w)
    return res


def main(fetch_db, write_db, out, workload="C2"):
    fetch, nf = per_kernel(fetch_db, "FETCH_SIZE")
    write, _ = per_kernel(write_db, "WRITE_SIZE")

    import os
    import subprocess
    import sys as _sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _sys.path.insert(0, repo)
    import bench
    try:
        commit = subprocess.run(["git", "-C", repo, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = ""
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on tools/prof_ops.py --ops agg,aggfd,fr (%s; fused walk: dense-prior launches only; unfused walk: dense and seed priors alternate)" % workload,
           "workload": workload, "commit": commit or os.environ.get("GNNRAG_COMMIT", "unknown (no .git on the GPU box)"),
           "kernel_sources_sha256": bench.kernel_sources_digest(),
           "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B)"}
    res.update(totals(fetch, write))
    res["per_kernel_fetch_KB"], res["per_kernel_write_KB"] = old["per_kernel_fetch_KB"], old["per_kernel_write_KB"]
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k.endswith("per_launch")}))


def totals(fetch, write):
    res = {}

    def group(pred, launches_per_call=1.0):
        f = sum(v for k, v in fetch.items() if pred(k)) * 1024.0
        w = sum(v for k, v in write.items() if pred(k)) * 1024.0
        return f, w

    f, w = group(lambda k: "k_walk_slice" in k or "k_fact_prior" in k or ("k_walk_light" in k and "ILi2E" in k)
                 or "k_walk_light_q" in k or ("k_heavy" in k and "ILi2E" in k) or "k_hub_" in k)
    res.update(aggregate_fused_fetch_bytes_raw=f, aggregate_fused_write_bytes=w,
               aggregate_fused_hbm_bytes_per_launch=2 * f + w)
    # the seed-prior (frontier) form of layer 0: frontier + table rows + neighbour sums of the frontier
    f, w = group(lambda k: "k_frontier_build" in k or "k_tables_frontier" in k or "k_walk_frontier" in k)
    res.update(frontier_fetch_bytes_raw=f, frontier_write_bytes=w, frontier_hbm_bytes_per_launch=2 * f + w)
    f, w = group(lambda k: ("k_walk_light" in k or "k_heavy" in k) and "ILi0E" in k)
    res.update(aggregate_fetch_bytes_raw=f, aggregate_write_bytes=w, aggregate_hbm_bytes_per_launch=2 * f + w)
    # the self-block update with a dense nbr (k_update_b3<false>); its row-gated form of layer 0 (k_update_b3<true>, rows off
    # the frontier read a zero row) is a different launch and is kept apart
    f, w = group(lambda k: ("k_update_b3ILb0E" in k))
    if f or w:
        res.update(update_score_fused_fetch_bytes_raw=f, update_score_fused_write_bytes=w,
                   update_score_fused_hbm_bytes_per_launch=2 * f + w)
    f, w = group(lambda k: ("k_update_b3ILb1E" in k))
    if f or w:
        res.update(update_score_gated_fetch_bytes_raw=f, update_score_gated_write_bytes=w,
                   update_score_gated_hbm_bytes_per_launch=2 * f + w)
    return res


def main(fetch_db, write_db, out, workload="C2"):
    fetch, nf = per_kernel(fetch_db, "FETCH_SIZE")
    write, _ = per_kernel(write_db, "WRITE_SIZE")

    import os
    import subprocess
    import sys as _sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _sys.path.insert(0, repo)
    import bench
    try:
        commit = subprocess.run(["git", "-C", repo, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = ""
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on tools/prof_ops.py --ops agg,aggfd,fr (%s; fused walk: dense-prior launches only; unfused walk: dense and seed priors alternate)" % workload,
           "workload": workload, "commit": commit or os.environ.get("GNNRAG_COMMIT", "unknown (no .git on the GPU box)"),
           "kernel_sources_sha256": bench.kernel_sources_digest(),
           "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B)"}
    res.update(totals(fetch, write))          # one grouping for fresh runs and --recompute
    res["per_kernel_fetch_KB"] = {k[:60]: v for k, v in fetch.items() if "gnnrag" in k}
    res["per_kernel_write_KB"] = {k[:60]: v for k, v in write.items() if "gnnrag" in k}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k.endswith("per_launch")}))


if __name__ == "__main__":
    if sys.argv[1] == "--recompute":
        for p in sys.argv[2:]:
            recompute(p)
    else:
        main(*sys.argv[1:5])
