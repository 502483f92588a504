#!/usr/bin/env python3
"""Per-kernel statistics (and optionally one step's timeline) from a rocprofv3 rocpd SQLite database (the default output
format of ROCm 7.2's `rocprofv3 --kernel-trace`).

    tools/rocpd_stats.py <out_results.db> [--csv profiles/<tag>_kernel_stats.csv] [--timeline <anchor kernel substring>]
"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[0-9, a-z]+>)?)", name)
    s = m.group(1) if m else name
    if s.startswith("rocprim"):
        k = re.search(r"(radix_sort_[a-z_]+|lookback_scan_[a-z_]+|init_lookback[a-z_]+|scan_[a-z_]+|trampoline_kernel)", name)
        s = "rocprim::" + (k.group(1) if k else "kernel")
    return s


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in cols else ("display_name" if "display_name" in cols else cols[-1])
    names = {r[0]: r[1] for r in cur.execute("select id, %s from %s" % (namecol, ks))}
    rows = [(names.get(k, str(k)), q, s, e) for k, q, s, e in cur.execute("select kernel_id, queue_id, start, end from %s order by start" % kd)]
    return rows


def main():
    rows = load(sys.argv[1])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, q, s, e in rows:
        a = agg[short(name)]
        a[0] += 1
        a[1] += (e - s) / 1e3
    total = sum(v[1] for v in agg.values())
    lines = ["kernel,calls,total_us,avg_us,percent"]
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%s,%d,%.1f,%.2f,%.2f" % (k, c, t, t / c, 100 * t / total))
    if "--csv" in sys.argv:
        with open(sys.argv[sys.argv.index("--csv") + 1], "w") as f:
            f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))
    if "--timeline" in sys.argv:
        anchor = sys.argv[sys.argv.index("--timeline") + 1]
        idx = [i for i, r in enumerate(rows) if anchor in r[0]]
        if len(idx) > 12:
            i0, i1 = idx[10], idx[11]
            t0 = rows[i0][2]
            print("\n-- one step (us since its first kernel: start, duration, queue, kernel)")
            for name, q, s, e in rows[i0:i1]:
                print("%8.1f %7.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short(name)))


if __name__ == "__main__":
    main()
