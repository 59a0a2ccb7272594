#!/bin/bash
# A second library beside the shipped one, for same-box A/B runs (CAPTRA_LIB=<out.so> python ...): the named sources recompiled with extra
# flags (e.g. -DCAPTRA_ABLATIONS=1), every other object taken from the regular build (run `python -m captra_amd.build` first).
# Usage: tools/build_variant.sh abtmp/libabl.so "-DCAPTRA_ABLATIONS=1" pointwise_mlp.hip [more.hip ...]
set -e
cd "$(dirname "$0")/.."
out=$1; extra=$2; shift 2
mkdir -p abtmp/obj
objs=""
for o in captra_amd/csrc/_obj/*.o; do
  base=$(basename "$o" .o); skip=0
  for s in "$@"; do [ "$s" = "$base" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for s in "$@"; do
  per=""
  case $s in sa_fused.hip|mlp_chain.hip|fps.hip|sa_bf16.hip) per="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=200000 -fno-fast-math \
    -Iinclude -Icaptra_amd/csrc $per $extra -x hip -c captra_amd/csrc/$s -o abtmp/obj/$s.o
  objs="$objs abtmp/obj/$s.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs
echo "built $out"
