#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (text).
usage: python tools/rocpd_summary.py results.db [> profiles/rNN_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    suf = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like "
                                    "'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         "max(d.end-d.start), max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size), "
         "max(d.private_segment_size), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count) "
         "from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id "
         "group by s.kernel_name order by 3 desc" % (suf, suf))
    rows = list(db.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("%-78s %7s %12s %10s %10s %10s %6s %8s %5s %8s %8s %5s %5s %5s" % (
        "kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "grid", "wg", "lds_B", "scratch", "vgpr",
        "agpr", "sgpr"))
    for r in rows:
        name = r[0] if len(r[0]) <= 78 else r[0][:75] + "..."
        print("%-78s %7d %12d %10.0f %10d %10d %6.2f %8d %5d %8d %8d %5d %5d %5d" % (
            name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))


if __name__ == "__main__":
    main(sys.argv[1])
