#!/bin/bash
# usage: tools/variants.sh "<bench args>" lib1 lib2 ...   (GPU box; compares builds of the library)
args=$1; shift
for l in "$@"; do
  MERCAT_HIP_LIB=$PWD/build/libmercat_$l.so timeout -k 5 200 python bench.py --no-cpu $args > gpurun_out/var_$l.log 2>&1
  echo "$l: $(grep -o '"value": [0-9.e+]*' gpurun_out/var_$l.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/var_$l.log) $(grep -o '"count": [0-9.]*' gpurun_out/var_$l.log) rows $(grep -o '"rows": [0-9]*' gpurun_out/var_$l.log)"
done
