#!/bin/bash
# Build an experimental libamvs variant for A/B runs:  tools/build_variant.sh NAME -DFOO=1 ...
# -> build/variants/libamvs_NAME.so   (select it at run time with AMVS_LIB=<path>)
# The flags apply to the kernel translation units; the other objects are reused -- ALL=1 rebuilds every
# translation unit with the flags (needed for -DAMVS_CHECK_INDICES, the index-checked build:
#     ALL=1 tools/build_variant.sh check -DAMVS_CHECK_INDICES;  AMVS_LIB=$PWD/build/variants/libamvs_check.so pytest tests -m gpu)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/3d-reconstruction-tool_amd/csrc
out=$root/build/variants; mkdir -p $out/obj_$name
flags="--offload-arch=${ARCH:-gfx950} -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-result"
for f in amvs_kernels amvs_kernels_fast amvs_sweep_fast amvs_sweep_exact amvs_generic amvs_capi amvs_fusion amvs_knn amvs_prep amvs_extended amvs_pool; do
  if [ $f = amvs_kernels_fast ] || [ $f = amvs_sweep_fast ] || { [ $f = amvs_sweep_exact ] && [ -z "$FAST_ONLY" ]; } || { [ $f = amvs_kernels ] && [ -z "$FAST_ONLY" ]; } || [ -n "$ALL" ] || [ ! -f $src/$f.o ]; then
    extra=""; { [ $f = amvs_sweep_fast ] || [ $f = amvs_sweep_exact ]; } && extra="-mllvm -amdgpu-sched-strategy=iterative-maxocc"    # as csrc/Makefile
    /opt/rocm/bin/hipcc $flags $extra "$@" -c $src/$f.hip -o $out/obj_$name/$f.o &
  else
    cp $src/$f.o $out/obj_$name/$f.o
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=${ARCH:-gfx950} -shared -fPIC -o $out/libamvs_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/libamvs_$name.so
