#!/usr/bin/env python
"""Copies the judged summaries of one tools/refresh_profiles.sh run from gpurun_out/<tag>/ into profiles/ (tracked).
    python tools/collect_profiles.py r02a"""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main(tag):
    src = os.path.join(REPO, "gpurun_out", tag)
    dst = os.path.join(REPO, "profiles")
    json.dump(last_json(os.path.join(src, "bench_default.log")), open(os.path.join(dst, tag + "_bench_default.json"), "w"), indent=1)
    json.dump(last_json(os.path.join(src, "bench_under_rocprof.log")), open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w"), indent=1)
    for f, g in (("kernel_stats_bench.txt", "_kernel_stats_bench.txt"), ("kernel_stats_bench_C5.txt", "_kernel_stats_bench_C5.txt"),
                 ("pmc_traffic_C2.json", "_pmc_traffic_C2.json"), ("pmc_traffic_C3.json", "_pmc_traffic_C3.json"),
                 ("pmc_traffic_C4.json", "_pmc_traffic_C4.json"), ("pmc_traffic_C5.json", "_pmc_traffic_C5.json")):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, tag + g))
    shutil.copy(os.path.join(src, "pmc_traffic_C2.json"), os.path.join(dst, "pmc_traffic.json"))
    for w in ("C3", "C4", "C5"):               # bench.py's other_workloads legs read these (one file per workload)
        if os.path.exists(os.path.join(src, "pmc_traffic_%s.json" % w)):
            shutil.copy(os.path.join(src, "pmc_traffic_%s.json" % w), os.path.join(dst, "pmc_traffic_%s.json" % w))
    other = {}
    for w in ("C1", "C3", "C4", "C5", "C2fb", "C2u", "C1_graph", "C3_graph"):
        p = os.path.join(src, "bench_%s.log" % w)
        if os.path.exists(p):
            try:
                other[w] = last_json(p)
            except Exception as e:
                other[w] = {"error": repr(e)}
    json.dump(other, open(os.path.join(dst, tag + "_other_workloads.json"), "w"), indent=1)
    print("profiles/%s_*: default bench %.4f ms/step; others: %s" % (
        tag, last_json(os.path.join(src, "bench_default.log"))["ms_per_step"],
        {k: round(v.get("ms_per_step", -1), 4) for k, v in other.items()}))


if __name__ == "__main__":
    main(sys.argv[1])
