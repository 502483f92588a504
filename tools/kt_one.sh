#!/bin/bash
# kernel table of one bench.py configuration:  tools/kt_one.sh <tag> [bench args ...]   (summary -> gpurun_out/<tag>_kt.txt)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kt -o out -- python3 bench.py --no-extra-legs --no-cpu-baseline --repeats 1 --steps 30 "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/kt.err
f=$(find gpurun_out/$tag/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_kt.txt <<'PY'
import csv,sys,re
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n)[:70]
    rows.append((float(r["TotalDurationNs"]),int(r["Calls"]),n))
rows.sort(reverse=True)
for t,c,n in rows[:40]: print("%9.1f us total  %6d calls  %8.2f us avg  %s"%(t/1e3,c,t/1e3/c,n))
PY
python3 -c "import json;d=json.loads(open('gpurun_out/$tag/bench.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])" >> gpurun_out/${tag}_kt.txt
python3 tools/step_timeline.py "$(find gpurun_out/$tag/kt -name "*kernel_trace.csv" | head -1)" > gpurun_out/${tag}_timeline.txt 2>&1
rm -rf gpurun_out/$tag/kt
cat gpurun_out/${tag}_kt.txt
