#!/bin/bash
# dev tool (GPU box): SQ occupancy / issue counters of the bench's kernels, one --pmc pass per counter group
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d /root/repo/gpurun_out/pmc_sq1 -o s1 --output-format csv -- python3 /root/repo/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/pmc_sq1.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM -d /root/repo/gpurun_out/pmc_sq2 -o s2 --output-format csv -- python3 /root/repo/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/pmc_sq2.log 2>&1
ls /root/repo/gpurun_out/pmc_sq1 /root/repo/gpurun_out/pmc_sq2; tail -2 /root/repo/gpurun_out/pmc_sq2.log | cut -c1-300
