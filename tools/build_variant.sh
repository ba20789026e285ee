#!/bin/bash
# usage: tools/build_variant.sh <tag> "<extra hipcc flags>" [file.hip ...]   (default file: mk_skcount.hip, the count kernels)
# Builds build/libmercat_<tag>.so: the named sources recompiled with the extra flags, every other object as in
# the product build.  For A/B runs on the GPU box: MERCAT_HIP_LIB=$PWD/build/libmercat_<tag>.so python bench.py ...
set -e
tag=$1; flags=$2; shift 2 || true
files=${@:-mk_skcount.hip}
cd "$(dirname "$0")/../mercat2_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../../build/obj_$tag
skip=""
for f in $files; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -I../../include $flags -c $f -o ../../build/obj_$tag/${f%.hip}.o
  skip="$skip ${f%.hip}.o"
done
objs=""
for o in ../../build/obj/*.o; do
  b=$(basename $o)
  case " $skip " in *" $b "*) objs="$objs ../../build/obj_$tag/$b";; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/libmercat_$tag.so $objs -lz -lpthread
echo "built build/libmercat_$tag.so"
