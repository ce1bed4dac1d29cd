#!/bin/bash
# Builds jxl_rs_amd/variants/libjxl_hip_<name>.so: the product library with ONE translation unit
# recompiled under extra -D flags (kernel tuning experiments; select with JXLH_LIBRARY=<path>).
#   tools/build_variant.sh <name> <file.hip> [-DFLAG=..]...
set -e
cd "$(dirname "$0")/../jxl_rs_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../variants
make -s -j8 >/dev/null
obj=../variants/${src%.hip}_$name.o
# the translation unit's own extra flags (Makefile: EXTRA_<name> := ...), e.g. -fno-slp-vectorize for the fused filters
extra=$(sed -n "s/^EXTRA_${src%.hip} *:= *//p" Makefile)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function $extra "$@" -c $src -o $obj 2>&1 | grep -v "hip-link\|^clang" || true
others=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libjxl_hip_$name.so $obj $others 2>&1 | grep -v "hip-link\|^clang" || true
rm -f $obj
echo "built jxl_rs_amd/variants/libjxl_hip_$name.so"
