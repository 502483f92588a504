#!/bin/bash
# diagnostics: A/B runs of bench.py with phase timers (gpurun_out/<tag>/sweep.txt)
out=$1; shift
mkdir -p $out
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python bench.py --no-extra-legs --no-cpu-baseline --profile-all "$@" > $out/$name.json 2> $out/$name.err || return 1
  python3 - "$name" "$out/$name.json" >> $out/sweep.txt <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
ph=d.get("phase_ms_per_step",{})
print("%-22s step %.4f  " % (sys.argv[1], d["ms_per_step"]) + " ".join("%s %.3f" % (k, v) for k, v in ph.items()) + "  frac %.3f" % d.get("roofline",{}).get("frac",0))
PY
}
