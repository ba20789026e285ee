#!/bin/bash
# usage: tools/ab_s1.sh <rounds> "tag[:ENV=1]" ...   (GPU box) -- interleaved A/B on the S1 workload (1 M reads, k = 21, BASELINE config 2)
rounds=$1; shift
declare -A best
for ((r = 0; r < rounds; ++r)); do
  for v in "$@"; do
    tag=${v%%:*}; envs=""; [ "$v" != "$tag" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    lib=$PWD/build/libmercat_$tag.so; [ $tag = hip ] && lib=$PWD/mercat2_amd/libmercat_hip.so
    out=$(env MERCAT_HIP_LIB=$lib $envs timeout -k 10 200 python bench.py --no-cpu --no-file-leg --no-configs --steps 10 --warmup 3 --reads 1000000 --genome 1000000 --k 21 --genome-seed 1 --read-seed 2 2>/dev/null)
    ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    cms=$(echo "$out" | grep -o '"ms_per_launch": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "round $r $v step_ms=$ms count_launch_ms=$cms"
    if [ -n "$ms" ] && { [ -z "${best[$v]}" ] || awk "BEGIN{exit !($ms < ${best[$v]})}"; }; then best[$v]=$ms; fi
  done
done
for v in "$@"; do echo "min $v ${best[$v]}"; done
