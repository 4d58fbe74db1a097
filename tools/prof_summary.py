#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite database or kernel_trace csv) per kernel:
calls, total/avg/max duration.  Usage: tools/prof_summary.py <results.db|kernel_trace.csv> [out.txt]"""
import csv
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, d.end-d.start, d.grid_size_x*d.grid_size_y*d.grid_size_z, d.workgroup_size_x, s.arch_vgpr_count, s.sgpr_count, d.group_segment_size "
         f"from {kd} d join {ks} s on d.kernel_id=s.id")
    return list(cur.execute(q))


def from_csv(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r.get("Grid_Size", 0) or 0),
                     int(r.get("Workgroup_Size", 0) or 0), int(r.get("VGPR_Count", 0) or 0), int(r.get("SGPR_Count", 0) or 0), int(r.get("LDS_Block_Size", 0) or 0)))
    return rows


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = defaultdict(lambda: [0, 0, 0, None])
    for name, dur, grid, wg, vgpr, sgpr, lds in rows:
        a = agg[name]
        a[0] += 1; a[1] += dur; a[2] = max(a[2], dur); a[3] = (grid, wg, vgpr, sgpr, lds)
    tot = sum(a[1] for a in agg.values())
    lines = [f"{'kernel':58s} {'calls':>5s} {'total_ms':>10s} {'avg_ms':>10s} {'max_ms':>10s} {'%':>6s}  grid wg vgpr sgpr lds"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name.replace(".kd", "")
        lines.append(f"{short[:58]:58s} {a[0]:5d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e6:10.3f} {a[2] / 1e6:10.3f} {100.0 * a[1] / tot:6.2f}  {a[3]}")
    lines.append(f"{'TOTAL':58s} {'':5s} {tot / 1e6:10.3f}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
