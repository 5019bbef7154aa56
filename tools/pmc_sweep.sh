#!/bin/bash
# GPU box: rocprofv3 PMC passes of the plane-sweep kernels (one counter set per run; --pmc alone, no trace domains).
# Usage: tools/pmc_sweep.sh <tag> <workload> [more workloads]; output gpurun_out/<tag>/pmc_summary.txt
tag=${1:-pmc}; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
for wl in "$@"; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/pmc_${wl}_$i -o pmc -- python3 tools/profile_sweep.py $wl 1 > $out/pmc_${wl}_$i.log 2>&1
  done
  echo "## $wl" >> $out/pmc_summary.txt
  for j in 1 2 3 4 5 6; do python3 tools/pmc_summary.py $out/pmc_${wl}_$j | grep "plane_sweep" >> $out/pmc_summary.txt; done
done
find $out -name "*.csv" -size +2000k -delete
cat $out/pmc_summary.txt
