#!/bin/bash
# dev tool (GPU box): round-3 measurement set on the tree as it stands -- VALU issue-rate microbenchmark, kernel trace,
# SQ counter passes (one --pmc run each, <= 8 SQ counters), HBM traffic passes.  usage: tools/r3_measure.sh <tag>
tag=${1:-r3}
R=/root/repo/gpurun_out
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-inference-extra"
timeout -k 10 120 /root/repo/tools/valu_issue_rate.bin > $R/${tag}_valu_issue_rate.txt 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/prof_$tag -o $tag --output-format csv -- $B > $R/prof_$tag.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $R/pmc_${tag}_sq1 -o s1 --output-format csv -- $B > $R/pmc_${tag}_sq1.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $R/pmc_${tag}_sq2 -o s2 --output-format csv -- $B > $R/pmc_${tag}_sq2.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_${tag}_fetch -o f --output-format csv -- $B > $R/pmc_${tag}_fetch.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $R/pmc_${tag}_write -o w --output-format csv -- $B > $R/pmc_${tag}_write.log 2>&1
rc=$?
cat $R/${tag}_valu_issue_rate.txt
tail -c 600 $R/prof_$tag.log
exit $rc
