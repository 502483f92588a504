#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small files kept under profiles/.

    tools/rocprof_summary.py <round-tag> <kernel_stats.csv> [<fetch counter_collection.csv> <write counter_collection.csv>]

Writes profiles/<tag>_kernel_stats.csv (per-kernel calls / total / average, short names) and, when the two PMC
passes are given, profiles/<tag>_hbm_traffic.json with the per-launch HBM bytes of every kernel:
FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide (16 B per
lane) streaming reads (MI355X_MICROARCH.md, section HBM), so the read side is given raw and doubled.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[0-9, a-z]+>)?)", name)
    s = m.group(1) if m else name
    if s.startswith("rocprim"):
        k = re.search(r"(radix_sort_[a-z_]+|lookback_scan_[a-z_]+|init_lookback[a-z_]+|scan_[a-z_]+|trampoline_kernel)", name)
        s = "rocprim::" + (k.group(1) if k else "kernel")
    return s


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    rows = defaultdict(lambda: [0, 0.0])
    with open(stats) as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            rows[k][0] += int(r["Calls"])
            rows[k][1] += float(r["TotalDurationNs"])
    total = sum(v[1] for v in rows.values())
    out = os.path.join(root, tag + "_kernel_stats.csv")
    with open(out, "w") as f:
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            f.write("%s,%d,%.1f,%.2f,%.2f\n" % (k, c, t / 1e3, t / c / 1e3, 100 * t / total))
    print("wrote", out)
    if len(sys.argv) >= 5:
        agg = {}
        for path, ctr in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            acc = defaultdict(lambda: [0, 0.0])
            with open(path) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] != ctr:
                        continue
                    k = short(r["Kernel_Name"])
                    acc[k][0] += 1
                    acc[k][1] += float(r["Counter_Value"])
            for k, (c, v) in acc.items():
                agg.setdefault(k, {})[ctr] = {"launches": c, "avg_kib": v / c}
        res = {}
        for k, d in agg.items():
            fe = d.get("FETCH_SIZE", {}).get("avg_kib", 0.0) * 1024
            wr = d.get("WRITE_SIZE", {}).get("avg_kib", 0.0) * 1024
            res[k] = {"fetch_bytes_raw": fe, "write_bytes": wr, "hbm_bytes_raw": fe + wr, "hbm_bytes_fetch_doubled": 2 * fe + wr,
                      "launches": d.get("FETCH_SIZE", d.get("WRITE_SIZE"))["launches"]}
        out = os.path.join(root, tag + "_hbm_traffic.json")
        with open(out, "w") as f:
            json.dump({"_what": "per-launch averages from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB -> bytes); "
                                "gfx950: FETCH_SIZE tallies 128-B read requests at 64 B, so wide streaming reads need the doubled figure",
                       "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_raw"]))}, f, indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()
