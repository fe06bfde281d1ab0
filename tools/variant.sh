#!/bin/bash
# Build an experimental variant of ONE translation unit and link it with the stock objects:
#   tools/variant.sh <tag> <unit> [-DFLAG ...]   ->  magellanmapper_amd/libmmx_<tag>.so  (select with MMX_LIB_PATH)
set -e
cd "$(dirname "$0")/../magellanmapper_amd/csrc"
tag=$1; unit=$2; shift 2
mkdir -p _obj_var
extra=""
case $unit in mmx_fused) extra="-mllvm -unroll-threshold=200000";; mmx_rescore|mmx_tables|mmx_preproc) extra="-ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 \
      -Wno-unused-value $extra "$@" -c $unit.hip -o _obj_var/${unit}_$tag.o
objs=$(ls _obj/*.o | grep -v "/$unit.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmmx_$tag.so $objs _obj_var/${unit}_$tag.o
echo built ../libmmx_$tag.so
