#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs (rocpd sqlite): per kernel and counter, launches, total and per-launch value.
Usage: tools/pmc_summary.py <results.db> [<results.db> ...]   (one database per --pmc pass)"""
import sqlite3
import sys
from collections import defaultdict


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        agg = defaultdict(lambda: [0, 0.0])
        for kname, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
            a = agg[(cname, kname)]
            a[0] += 1; a[1] += val
        for (cname, kname), (n, tot) in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
            if tot <= 0:
                continue
            print(f"{cname:11s} {kname.replace('.kd', '')[:64]:64s} n={n:4d} total={tot:16.0f} per_launch={tot / n:16.0f}")


if __name__ == "__main__":
    main()
