#!/bin/bash
# tools/dev/build_variant.sh NAME "-DFLAG ..." [files...]: an experimental build of the library with extra macros in the
# given translation units (default: bundle.hip), linked with the product objects of the rest -> tools/_exp/NAME/libptam_hip.so
# (used via PTAM_HIP_LIB; never shipped as the product)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; FLAGS=$2; shift 2
FILES=${@:-bundle.hip}
C=$R/ptam_cg_amd/csrc
O=$R/tools/_exp/$NAME
mkdir -p $O
make -s -C $C -j8
OBJS=""
for f in ctx keyframe patch pose pvs trackmap motion bundle solve comm; do
  if echo " $FILES " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -Wno-unused-value -Wno-unused-result $FLAGS -c $C/$f.hip -o $O/$f.o
    OBJS="$OBJS $O/$f.o"
  else
    OBJS="$OBJS $C/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libptam_hip.so $OBJS -ldl -Wl,-rpath,/opt/rocm/lib
echo built $O/libptam_hip.so
