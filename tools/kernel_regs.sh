#!/bin/bash
# dev tool: VGPR / SGPR / LDS / scratch of gsx_composite.hip's kernels matching <pattern> under extra -D flags (device-only -S).
# usage: tools/kernel_regs.sh <pattern> [-DX=Y ...]
pat=$1; shift
cd "$(dirname "$0")/../gaussian-splatting_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=fast -fno-fast-math -fno-slp-vectorize "$@" -S --cuda-device-only ${SRC:-gsx_composite.hip} -o /tmp/kregs.s 2>/dev/null
python3 - "$pat" <<'PY'
import re, sys
s = open('/tmp/kregs.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    if re.search(sys.argv[1], m.group(1)):
        g = lambda k: re.search(k + r' (\S+)', m.group(2)).group(1)
        print(m.group(1)[:60], 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'))
PY
