#!/bin/bash
# usage: tools/pmc_sq.sh <tag>      (GPU box, repo root)
# SQ counter passes (8 slots per pass on gfx950) over one context counting S2-shaped chunks (tools/parse_probe.py):
# instruction mix, issue / wait split and LDS conflict cycles of every kernel, for the table in profiles/.
tag=${1:-sq}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C="SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
timeout -k 5 240 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_A -- python3 $root/tools/parse_probe.py 4 > $root/gpurun_out/${tag}_A.log 2>&1 &&
timeout -k 5 240 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_B -- python3 $root/tools/parse_probe.py 4 > $root/gpurun_out/${tag}_B.log 2>&1 &&
timeout -k 5 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_C -- python3 $root/tools/parse_probe.py 4 > $root/gpurun_out/${tag}_C.log 2>&1
echo "pmc_sq rc=$?"
