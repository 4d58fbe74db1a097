#!/usr/bin/env python3
"""Where the GPU idles: from a rocprofv3 --kernel-trace database, the dispatches in time order (k_rc and the synthetic-input kernels
left out): busy time (union of the kernels' intervals), the idle gaps between them and what stands on either side of the longest.
Usage: tools/prof_timeline.py <results.db> [first_ms last_ms]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    t0 = rows[0][1]
    rows = [(n.split("(")[0].replace(".kd", "")[:40], (a - t0) / 1e6, (b - t0) / 1e6) for n, a, b in rows]
    lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1e18)
    fe = [r for r in rows if "k_rc" != r[0][:4] and "k_synth" not in r[0] and "lds_order" not in r[0] and "selftest" not in r[0] and lo <= r[1] <= hi]
    rc = [r for r in rows if r[0].startswith("_Z4k_rc") and lo <= r[1] <= hi]
    print("k_rc launches:", [(round(a, 1), round(b, 1)) for _, a, b in rc])
    if not fe:
        return
    busy = 0.0; cur_a, cur_b = fe[0][1], fe[0][2]; gaps = []; last = fe[0]
    for r in fe[1:]:
        if r[1] > cur_b:
            busy += cur_b - cur_a; gaps.append((r[1] - cur_b, cur_b, last[0], r[0])); cur_a, cur_b = r[1], r[2]
        else:
            cur_b = max(cur_b, r[2])
        if r[2] >= cur_b: last = r
    busy += cur_b - cur_a
    span = fe[-1][2] - fe[0][1]
    print(f"{len(fe)} dispatches over {span:.1f} ms: busy {busy:.1f} ms, idle {span - busy:.1f} ms in {len(gaps)} gaps; sum of durations {sum(b - a for _, a, b in fe):.1f} ms")
    for g in sorted(gaps, reverse=True)[:25]:
        print(f"  {g[0]:8.3f} ms idle at {g[1]:9.1f}: after {g[2]:40s} before {g[3]}")
    small = [g for g in gaps if g[0] < 0.05]
    print(f"  gaps under 50 us: {len(small)}, {sum(g[0] for g in small):.2f} ms in all")


if __name__ == "__main__":
    main()
