#!/usr/bin/env python
"""Kernel timeline of bench steps from a rocprofv3 rocpd database (``--kernel-trace``): start / end / duration of every
launch of one step in the middle of the timed loop, relative to the step's first launch, with the hardware queue it ran
on - what overlaps what when the whole-iteration call forks its side stream.
Usage: python tools/step_timeline.py results.db [step index, default: the middle one] [steps to print, default 1]"""
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"gnnrag\d+(k_[a-z_0-9]+)", n)
    return m.group(1) if m else n[:30]


def main(path, which=None, count=1):
    c = sqlite3.connect(path)
    rows = c.execute("""select s.kernel_name, d.start, d.end, d.queue_id from rocpd_kernel_dispatch d
                        join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    # a step starts with the relation projections followed (within two launches) by the frontier build
    idx = [i for i, r in enumerate(rows[:-2]) if "rel_transform" in r[0] and any("frontier_build" in x[0] for x in rows[i + 1:i + 3])]
    if len(idx) < 3:
        raise SystemExit("no steps found")
    k = len(idx) // 2 if which is None else which
    i0, i1 = idx[k], idx[min(k + count, len(idx) - 1)]
    t0 = rows[i0][1]
    print("step %d of %d (launch order; us relative to the step's first launch)" % (k, len(idx)))
    for r in rows[i0:i1]:
        print("%-22s start %8.1f  end %8.1f  dur %7.1f  queue %s" % (short(r[0]), (r[1] - t0) / 1e3, (r[2] - t0) / 1e3,
                                                                    (r[2] - r[1]) / 1e3, r[3]))
    spans = [rows[idx[j + 1]][1] - rows[idx[j]][1] for j in range(len(idx) // 4, 3 * len(idx) // 4)]
    print("step-to-step distance, middle half of the steps: mean %.1f us, min %.1f, max %.1f" % (
        sum(spans) / len(spans) / 1e3, min(spans) / 1e3, max(spans) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
