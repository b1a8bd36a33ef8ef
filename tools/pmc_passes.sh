#!/bin/bash
# dev tool (GPU box): the two --pmc passes behind profiles/r1_pmc_hbm_traffic.json, each in a run of its own
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_fetch -o f --output-format csv -- python3 /root/repo/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/pmc_fetch.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/pmc_write -o w --output-format csv -- python3 /root/repo/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/pmc_write.log 2>&1
ls /root/repo/gpurun_out/pmc_fetch /root/repo/gpurun_out/pmc_write
