#!/usr/bin/env python3
"""Per-kernel SQ counter averages from a rocprofv3 --pmc counter_collection.csv (steady-state launches: the first `skip`
launches of every kernel are dropped).

    tools/sq_summary.py <counter_collection.csv> <out.json> [kernel-substring ...]

Derived (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles):
    dur_us                     from the dispatch timestamps
    valu_insts_per_wave        SQ_INSTS_VALU / SQ_WAVES
    valu_active_frac           SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (share of a wave's lifetime spent issuing vector instructions)
    wait_any_frac              SQ_WAIT_ANY / SQ_WAVE_CYCLES           (parked on s_waitcnt / barriers)
    wait_inst_frac             SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES      (issue stalls)
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[0-9, a-z]+>)?)", name)
    return m.group(1) if m else name


def main():
    path, out = sys.argv[1], sys.argv[2]
    want = sys.argv[3:]
    per = defaultdict(lambda: defaultdict(dict))     # kernel -> dispatch -> counter -> value
    dur = defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if want and not any(w in k for w in want):
                continue
            d = int(r["Dispatch_Id"])
            per[k][d][r["Counter_Name"]] = per[k][d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            dur[k][d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    res = {}
    for k, disp in per.items():
        ids = sorted(disp)
        skip = min(3, max(0, len(ids) - 1))
        ids = ids[skip:]
        avg = defaultdict(float)
        for d in ids:
            for c, v in disp[d].items():
                avg[c] += v / len(ids)
        e = dict(avg)
        e["dur_us"] = sum(dur[k][d] for d in ids) / len(ids)
        e["launches_averaged"] = len(ids)
        wc = e.get("SQ_WAVE_CYCLES")
        if e.get("SQ_WAVES"):
            if "SQ_INSTS_VALU" in e:
                e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
        if wc:
            for name, key in (("SQ_ACTIVE_INST_VALU", "valu_active_frac"), ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac")):
                if name in e:
                    e[key] = e[name] / wc
        res[k] = e
    with open(out, "w") as f:
        json.dump({"_what": __doc__.strip().split("\n\n")[0] + "  Source: " + path, "kernels": res}, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
