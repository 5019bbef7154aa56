#!/bin/bash
# GPU box: the sweep's LDS box swizzle variants (build_ab/libS<v>.so, MVS_BOX_SWZ=<v>: a what-if macro of that experiment, not in the tree any more): kernel time (tools/profile_sweep.py) and the LDS
# conflict counters of one rocprofv3 --pmc pass each.  bash tools/study/r06_swizzle_ab.sh "0 1 2 3 4 5" [workload]
root=${GRAFT_REPO_ROOT:-$PWD}; wl=${2:-scannet_40v_64d_120x160}; export TMPDIR=/tmp; cd $root
for r in 1 2; do for v in $1; do echo -n "S$v: "; MVSDET_HIP_LIB=$root/build_ab/libS$v.so python3 tools/profile_sweep.py $wl 4 2>/dev/null | tail -1 | cut -c1-200; done; done
for v in $1; do
  MVSDET_HIP_LIB=$root/build_ab/libS$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $root/gpurun_out/swz_$v -o pmc -- python3 tools/profile_sweep.py $wl 1 > /dev/null 2>&1
  echo "S$v counters:"; python3 tools/pmc_summary.py $root/gpurun_out/swz_$v | grep "plane_sweep_variance" | awk '{print "   ", $(NF-2), $NF}'
  find $root/gpurun_out/swz_$v -name "*.csv" -size +500k -delete
done
