#!/bin/bash
# PMC passes (one counter set per run) of the fp16-storage stress workload (BASELINE configs[4] as worded), one 10-view chunk.
tag=${1:-r03_fp16}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
wl=stress_100v_128d_240x320_c256_f16
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 tools/profile_sweep.py $wl 2 > $out/kt.log 2>&1
find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/${wl}_kernel_stats.csv \;
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -o pmc -- python3 tools/profile_sweep.py $wl 1 > $out/pmc_$i.log 2>&1
done
echo "## $wl" > $out/pmc_summary.txt
for j in 1 2 3 4; do python3 tools/pmc_summary.py $out/pmc_$j | grep "plane_sweep\|pack_features" >> $out/pmc_summary.txt; done
find $out -name "*.csv" -size +2000k -delete
find $out -name "*_kernel_trace.csv" -delete
cat $out/pmc_summary.txt | cut -c1-150
