#!/bin/bash
# (build the diagnostic programs first, here or on the box: make -C tools)
# The round's profile set on the GPU box: kernel trace + stats, two HBM PMC passes, one SQ PMC pass (default and mutual near
# field), the pair-body ceiling, the default bench line.  Everything lands under gpurun_out/<tag>/ ; tools/rocprof_summary.py and
# tools/sq_summary.py condense it into profiles/<tag>_* afterwards (on either machine).
#   tools/profile_round.sh <tag>
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
B="bench.py --no-extra-legs --no-cpu-baseline --steps 30"
echo "[$(date +%T)] kernel trace" | tee -a $out/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o out -- python3 $B > $out/bench_under_rocprof.json 2> $out/kt.err || exit 2
echo "[$(date +%T)] FETCH_SIZE" | tee -a $out/progress.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o out -- python3 $B > /dev/null 2> $out/fetch.err || exit 3
echo "[$(date +%T)] WRITE_SIZE" | tee -a $out/progress.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o out -- python3 $B > /dev/null 2> $out/write.err || exit 4
echo "[$(date +%T)] SQ counters" | tee -a $out/progress.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/sq -o out -- python3 $B > /dev/null 2> $out/sq.err || exit 5
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/sq_mutual -o out -- python3 $B --engine-opt p2p_mutual=1 > /dev/null 2> $out/sq_mutual.err || exit 6
echo "[$(date +%T)] pair ceiling, clock / issue probe, near-field timeline" | tee -a $out/progress.txt
timeout -k 10 200 ./build/pair_ceiling $out/pair_ceiling.json > $out/pair_ceiling.txt 2>&1 || exit 7
timeout -k 10 200 ./build/valu_probe $out/valu_probe.json 5 > $out/valu_probe.txt 2>&1 || exit 7
NBCO_P2P_DUMP=/tmp/p2p_$tag.bin timeout -k 10 200 python3 tools/p2p_dump.py > $out/p2p_dump.txt 2>&1 && timeout -k 10 200 ./build/p2p_lab /tmp/p2p_$tag.bin $out/p2p_lab.json > $out/p2p_lab.txt 2>&1 || exit 7
echo "[$(date +%T)] default bench" | tee -a $out/progress.txt
timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 8
# condense on the box (the raw traces exceed what gpurun copies back), keep the summaries, drop the raw files
echo "[$(date +%T)] summaries" | tee -a $out/progress.txt
f() { find $out/$1 -name "$2" | head -1; }
python3 tools/rocprof_summary.py $tag "$(f kt '*kernel_stats.csv')" "$(f fetch '*counter_collection.csv')" "$(f write '*counter_collection.csv')" >> $out/progress.txt 2>&1
python3 tools/sq_summary.py "$(f sq '*counter_collection.csv')" profiles/${tag}_sq_counters.json >> $out/progress.txt 2>&1
python3 tools/sq_summary.py "$(f sq_mutual '*counter_collection.csv')" profiles/${tag}_sq_counters_mutual.json p2p l2p >> $out/progress.txt 2>&1
mkdir -p $out/summary
cp profiles/${tag}_* $out/summary/ 2>/dev/null
cp $out/bench_default.json $out/summary/${tag}_bench_default.json
cp $out/bench_under_rocprof.json $out/summary/${tag}_bench_under_rocprof.json
cp $out/pair_ceiling.json $out/summary/${tag}_pair_ceiling.json
cp $out/valu_probe.json $out/summary/${tag}_valu_probe.json
cp $out/p2p_lab.txt $out/summary/${tag}_p2p_lab.txt
cp $out/p2p_lab.json $out/summary/${tag}_p2p_lab.json
rm -rf $out/kt $out/fetch $out/write $out/sq $out/sq_mutual
echo "[$(date +%T)] done" | tee -a $out/progress.txt
ls -la $out/summary
