#!/bin/bash
# usage: tools/isa_count.sh [extra hipcc flags]  -> /tmp/isa/cnt_only.s : mk_sk_count_k<false,false> and <true,false> alone (20 s instead of 4 min)
mkdir -p /tmp/isa
cd "$(dirname "$0")/../mercat2_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -I../../include "$@" -S --cuda-device-only -o /tmp/isa/cnt_only.s mk_skcount.hip
python3 - <<'PY'
import re
s=open('/tmp/isa/cnt_only.s').read()
for m in re.finditer(r'\.name:\s+(_Z13mk_sk_count_k\w+)\n(?:.*\n){0,12}?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n){0,6}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', s):
    print(m.group(1)[:40], 'sgpr', m.group(2), 'spill', m.group(3), 'vgpr', m.group(4), 'spill', m.group(5))
PY
