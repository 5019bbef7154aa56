#!/bin/bash
# Same-box comparison of library builds (boxes differ by 3-5 %, more than most single changes): build the variants HERE into
# build_ab/lib<NAME>.so (an untracked directory that travels with the snapshot), then on the GPU box
#     bash tools/ab_libs.sh "A B C" "python tools/profile_sweep.py scannet_40v_64d_120x160 8" [rounds]
# runs the command once per library and round, alternating the libraries (MVSDET_HIP_LIB picks the .so: mvsdet_amd/_lib.py).
# How a variant is built without touching the tree's own library, e.g. the sweep with a macro set:
#     (cd mvsdet_amd/csrc && hipcc $FLAGS -DMVS_X=1 -c planesweep.hip -o /tmp/v.o && \
#      hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/libB.so api.o /tmp/v.o <the other objects>)
libs=${1:?names}
cmd=${2:?command}
rounds=${3:-2}
root=${GRAFT_REPO_ROOT:-$PWD}
for r in $(seq 1 $rounds); do
  for v in $libs; do
    echo -n "$v: "
    MVSDET_HIP_LIB=$root/build_ab/lib$v.so PYTHONPATH=$root timeout -k 10 300 $cmd 2>/dev/null | tail -1 | cut -c1-300
  done
done
