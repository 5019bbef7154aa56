#!/bin/bash
# counters of the per-tile and the persistent transposed kernel at conv11 (40 views): matrix pipes, waits, LDS
tag=${1:-r06ctp}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc1 -o pmc -- python3 tools/study/r06_convt_persist_pmc.py > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $out/pmc2 -o pmc -- python3 tools/study/r06_convt_persist_pmc.py > $out/pmc2.log 2>&1
# the what-if form (skip-tensor loads and stores beyond the descriptors): a library built with -DMVS_CONVT_WHATIF=3 under build_ab/libCTPW3.so, if present
if [ -f $root/build_ab/libCTPW3.so ]; then
  MVSDET_HIP_LIB=$root/build_ab/libCTPW3.so rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc3 -o pmc -- python3 tools/study/r06_convt_persist_pmc.py > $out/pmc3.log 2>&1
fi
echo "## counters, whole launch (3 dispatches each)" > $out/summary.txt
python3 tools/pmc_summary.py $out/pmc1 | grep "convT" >> $out/summary.txt
python3 tools/pmc_summary.py $out/pmc2 | grep "convT" >> $out/summary.txt
echo "## the persistent kernel with its skip-tensor loads and its stores sent beyond the descriptors (built with -DMVS_CONVT_WHATIF=3)" >> $out/summary.txt
python3 tools/pmc_summary.py $out/pmc3 | grep "convT" >> $out/summary.txt
find $out -name "*.csv" -size +2000k -delete
cat $out/summary.txt | cut -c1-200
