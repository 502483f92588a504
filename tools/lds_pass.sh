#!/bin/bash
# LDS counters of the step's kernels (one rocprofv3 --pmc pass of their own): tools/lds_pass.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_]*LDS[A-Z_]*|SQ_INST_CYCLES_[A-Z_]*|SQ_ACTIVE_INST_[A-Z_]*|SQ_INSTS_[A-Z_]*" | sort -u > $out/avail.txt
B="bench.py --no-extra-legs --no-cpu-baseline --steps 30 --repeats 1"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out/lds -o out -- python3 $B > /dev/null 2> $out/lds.err || { tail -5 $out/lds.err; exit 5; }
f=$(find $out/lds -name '*counter_collection.csv' | head -1)
python3 tools/sq_summary.py "$f" $out/lds_counters.json p2p m2l l2p subtree segsort >> $out/progress.txt 2>&1
rm -rf $out/lds
cat $out/progress.txt | tail -3
