cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
B="bench.py --no-extra-legs --no-cpu-baseline --repeats 1 --steps 30"
NBCO_LIB=build/libnbco_base.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c/kt_base -o out -- python3 $B > /dev/null 2> gpurun_out/r3c/kt_base.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c/kt_new -o out -- python3 $B > /dev/null 2> gpurun_out/r3c/kt_new.err
for t in base new; do f=$(find gpurun_out/r3c/kt_$t -name "*kernel_stats.csv" | head -1); echo "== $t $f"; python3 - "$f" <<'PY'
import csv,sys,re
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n)[:60]
    rows.append((float(r["TotalDurationNs"]),int(r["Calls"]),n))
rows.sort(reverse=True)
for t,c,n in rows[:32]: print("%9.1f us total  %6d calls  %8.2f us avg  %s"%(t/1e3,c,t/1e3/c,n))
PY
done
