#!/usr/bin/env python
"""Per-kernel statistics (calls, total/avg/min/max duration) from a rocprofv3 rocpd
sqlite database (``rocprofv3 --kernel-trace --stats -d DIR -o NAME`` writes NAME_results.db).
Usage: python tools/rocpd_stats.py results.db [> profiles/xxx_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start),
                  min(d.end - d.start), max(d.end - d.start), max(s.arch_vgpr_count), max(d.group_segment_size)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = c.execute(q).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-90s %7s %12s %10s %10s %10s %6s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                       "pct", "vgpr", "lds"))
    for name, n, s, a, mn, mx, vg, lds in rows:
        short = name if len(name) <= 90 else name[:87] + "..."
        print("%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %7s" % (short, n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3,
                                                                     100.0 * s / tot, vg, lds))


if __name__ == "__main__":
    main(sys.argv[1])
