#!/usr/bin/env python
"""Per-kernel averages of PMC counters from rocprofv3 rocpd sqlite databases.
Usage: python tools/rocpd_pmc.py a_results.db [b_results.db ...]"""
import sqlite3
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for p in paths:
        c = sqlite3.connect(p)
        q = """select s.kernel_name, d.id, i.name, sum(e.value), d.end - d.start
               from rocpd_pmc_event e join rocpd_info_pmc i on e.pmc_id = i.id
               join rocpd_kernel_dispatch d on e.event_id = d.event_id
               join rocpd_info_kernel_symbol s on d.kernel_id = s.id
               group by d.id, i.name"""
        try:
            rows = c.execute(q).fetchall()
        except sqlite3.OperationalError as ex:
            print("cannot read", p, ex)
            continue
        for name, did, cname, val, dt in rows:
            acc[name][cname].append(val)
            dur[name].append(dt)
    for name in sorted(acc, key=lambda n: -sum(dur[n])):
        short = name if len(name) < 100 else name[:97] + "..."
        print(short)
        for cname, vals in sorted(acc[name].items()):
            print("    %-28s n=%-4d avg=%-16.1f min=%-16.1f max=%-16.1f" % (cname, len(vals), sum(vals) / len(vals),
                                                                          min(vals), max(vals)))


if __name__ == "__main__":
    main(sys.argv[1:])
