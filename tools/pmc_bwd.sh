#!/bin/bash
# PMC passes of the backward sweep (tools/profile_bwd.py runs it four times at the reference-true shape).  $1 = tag under
# gpurun_out/; MVSDET_BWD_FIXED in the environment chooses the form of the LDS gradient images.
tag=${1:-r05_bwd_pmc}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$root
cd $root
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
           "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -o pmc -- python3 tools/profile_bwd.py > $out/pmc_$i.log 2>&1
  echo pass $i done
done
for j in 1 2 3 4 5; do python3 tools/pmc_summary.py $out/pmc_$j | grep "variance_bwd" >> $out/pmc_summary.txt; done
find $out -name "*.csv" -size +2000k -delete
cut -c1-200 $out/pmc_summary.txt
