#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace of the bench command + rocprofv3 PMC passes (one counter set per run,
# --pmc never combined with other trace domains) of tools/profile_sweep.py for the four BASELINE workloads.
# Outputs under gpurun_out/$1/; copy the summaries into profiles/ afterwards.
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --cpu-seconds 0 > $out/bench_under_rocprof.log 2>&1
find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_bwd -o kt -- python3 tools/profile_bwd.py > $out/kt_bwd.log 2>&1
find $out/kt_bwd -name "*kernel_stats.csv" -exec cp {} $out/bwd_scannet_ref_40v_12d_60x80_kernel_stats.csv \;
for wl in scannet_40v_64d_120x160 scannet_ref_40v_12d_60x80 arkit_50v_96d_60x80 stress_100v_128d_240x320_c32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$wl -o kt -- python3 tools/profile_sweep.py $wl 3 > $out/kt_$wl.log 2>&1
  find $out/kt_$wl -name "*kernel_stats.csv" -exec cp {} $out/${wl}_kernel_stats.csv \;
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/pmc_${wl}_$i -o pmc -- python3 tools/profile_sweep.py $wl 1 > $out/pmc_${wl}_$i.log 2>&1
  done
  echo "## $wl" >> $out/pmc_summary.txt
  for j in 1 2 3 4; do python3 tools/pmc_summary.py $out/pmc_${wl}_$j | grep "plane_sweep\|pack_features" >> $out/pmc_summary.txt; done
done
find $out -name "*.csv" -size +2000k -delete   # keep the merge small
find $out -name "*_kernel_trace.csv" -delete
ls $out
# every kernel of the hot path (pack, geometry, sweep, depth distribution, lifting) under the counters: the bench command itself
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/pmc_bench_$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-extras --cpu-seconds 0 > $out/pmc_bench_$i.log 2>&1
done
echo "## bench.py --steps 2 (scannet_40v_64d_120x160): all hot-path kernels" > $out/pmc_stage_summary.txt
for j in 1 2 3; do python3 tools/pmc_summary.py $out/pmc_bench_$j | grep "mvsdet::" >> $out/pmc_stage_summary.txt; done
find $out -name "*.csv" -size +2000k -delete
