"""Alternating A/B runs of bench.py under different environments on ONE box (the boxes of the pool differ by ~1 %):
    python tools/ab_env.py [--rounds 3] [--bench-args "..."] "label=ENV1=v;ENV2=w" "label2=" ...
Each variant is run once per round, in turn; prints ms_per_step (median repeat), the near-field launch time and the tree-reuse figure."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, bench_args = 3, "--no-cpu-baseline --repeats 3"
    while args and args[0].startswith("--"):
        if args[0] == "--rounds":
            rounds = int(args[1]); args = args[2:]
        elif args[0] == "--bench-args":
            bench_args = args[1]; args = args[2:]
        else:
            raise SystemExit("unknown option " + args[0])
    variants = []
    for a in args:
        label, _, envs = a.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(";") if kv) if envs else {}
        variants.append((label, env))
    res = {label: [] for label, _ in variants}
    for r in range(rounds):
        for label, env in variants:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + bench_args.split(), capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
            if p.returncode != 0:
                print(label, "FAILED", p.stderr[-300:])
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            row = (d["ms_per_step"], (d.get("roofline") or {}).get("avg_launch_ms"), (d.get("tree_reuse") or {}).get("ms_per_step"),
                   (d.get("near_field_mutual") or {}).get("ms_per_step"))
            res[label].append(row)
            print("round %d %-24s ms/step %.4f  near-field launch %s ms  tree_reuse %s  mutual %s" % (r, label, row[0], row[1], row[2], row[3]), flush=True)
    print("---- medians")
    for label, rows in res.items():
        if rows:
            med = lambda k: sorted(x[k] for x in rows if x[k] is not None)[len([x for x in rows if x[k] is not None]) // 2] if any(x[k] is not None for x in rows) else None
            print("%-24s ms/step %.4f  near-field launch %s  tree_reuse %s  mutual %s" % (label, med(0), med(1), med(2), med(3)))


if __name__ == "__main__":
    main()
