#!/bin/bash
# build a timing-only ablation of the fast kernels from a patched scratch copy (never committed)
# usage: mk.sh NAME 'python-replace-script'
set -e
name=$1; root=/root/repo; src=$root/3d-reconstruction-tool_amd/csrc
d=$root/build/exp/$name; rm -rf $d; mkdir -p $d
cp $src/*.h $src/amvs_kernels_fast.hip $d/
mkdir -p $d/../../include_tmp
python3 - "$d/amvs_kernels_fast.hip" <<PY
import sys
p=sys.argv[1]; s=open(p).read()
$2
open(p,"w").write(s)
PY
sed -i 's#"../../include/amvs.h"#"/root/repo/include/amvs.h"#' $d/*.h $d/*.hip
flags="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -Wno-unused-result "
/opt/rocm/bin/hipcc $flags -c $d/amvs_kernels_fast.hip -o $d/fast.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/variants/libamvs_$name.so $d/fast.o $src/amvs_kernels.o $src/amvs_capi.o $src/amvs_fusion.o $src/amvs_knn.o
echo built $name
