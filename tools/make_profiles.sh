#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 summaries of the benchmark command.
#   1. --kernel-trace --stats of `python3 bench.py --steps 3 --warmup 1 --no-cpu` (default: 2 contexts)
#   2. the same with --contexts 1 (kernel durations without cross-stream contention)
#   3. PMC passes (FETCH_SIZE, WRITE_SIZE separately; kernel-trace only) with --contexts 1
root=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp
for ctx in 2 1; do
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_ctx$ctx -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --no-file-leg --no-configs --contexts $ctx > $root/gpurun_out/prof_${tag}_ctx$ctx.log 2>&1 || exit 1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_$c -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu --no-file-leg --no-configs --contexts 1 > $root/gpurun_out/pmc_${tag}_$c.log 2>&1 || exit 1
done
# two-word keys (BASELINE config 5 shape: S3, k = 63), kernel stats only
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_k63 -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu --no-file-leg --k 63 --reads 50000000 --genome 50000000 --genome-seed 6 --read-seed 7 > $root/gpurun_out/prof_${tag}_k63.log 2>&1 || exit 1
# ... and their HBM traffic (PMC passes, kernel trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_k63_$c -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu --no-file-leg --k 63 --reads 50000000 --genome 50000000 --genome-seed 6 --read-seed 7 > $root/gpurun_out/pmc_${tag}_k63_$c.log 2>&1 || exit 1
done
echo done
