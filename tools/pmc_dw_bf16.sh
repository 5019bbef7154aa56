#!/bin/bash
# PMC passes of a bf16x3 weight-gradient kernel (tools/dw_bf16_one.py runs it a few times): $2 empty = stride 1 at the conv0 shape,
# $2 = s2: the stride-2 / transposed kernel at the conv1 shape.
tag=${1:-r03_dw}
which=${2:-}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$root
cd $root
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -o pmc -- python3 tools/dw_bf16_one.py $which > $out/pmc_$i.log 2>&1
  echo pass $i done
done
for j in 1 2 3 4 5; do python3 tools/pmc_summary.py $out/pmc_$j | grep "dw_bf16x3" >> $out/pmc_summary.txt; done
find $out -name "*.csv" -size +2000k -delete
cat $out/pmc_summary.txt | cut -c1-400
