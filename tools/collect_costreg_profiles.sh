#!/bin/bash
# rocprofv3 kernel trace + one PMC pass (matrix-core counters) of the cost network on our kernels (GPU box, gpurun).
tag=${1:-costreg}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o costreg -- python3 tools/costreg_layers_hip.py > $out/kt.log 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU_MFMA_MOPS_[A-Z0-9]*" | sort -u > $out/mfma_counters.txt
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/pmc -o pmc -- python3 tools/costreg_layers_hip.py scannet_ref_40v_12d_60x80 3 > $out/pmc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out/pmc2 -o pmc -- python3 tools/costreg_layers_hip.py scannet_ref_40v_12d_60x80 3 > $out/pmc2.log 2>&1
# round 6: conv0 runs on fp16 + block-scaled FP6 (csrc/costreg_mx.h): the matrix-op counters of those formats, whichever this rocprofv3 knows
extra=$(grep -E "MOPS_(F16|F8|F6F4|F6|F4)$" $out/mfma_counters.txt | tr '\n' ' ')
rocprofv3 --pmc $extra SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc3 -o pmc -- python3 tools/costreg_layers_hip.py scannet_ref_40v_12d_60x80 3 > $out/pmc3.log 2>&1
find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# the 3-D neck (eval) and one training step of the cost network (forward + backward on our kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_neck -o neck -- python3 tools/neck_timing.py > $out/kt_neck.log 2>&1
find $out/kt_neck -name "*kernel_stats.csv" -exec cp {} $out/neck_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_train -o train -- python3 tools/costreg_train_profile.py > $out/kt_train.log 2>&1
find $out/kt_train -name "*kernel_stats.csv" -exec cp {} $out/costreg_train_kernel_stats.csv \;
# matrix-core counters of the training step's kernels (forward, dX, the stride-1 and stride-2 / transposed weight gradients)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/pmc_train -o pmc -- python3 tools/costreg_train_profile.py > $out/pmc_train.log 2>&1
python3 tools/pmc_summary.py $out/pmc_train > $out/pmc_train_summary.txt 2>&1
find $out -name "*_kernel_trace.csv" -delete
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt 2>&1
python3 tools/pmc_summary.py $out/pmc2 >> $out/pmc_summary.txt 2>&1
python3 tools/pmc_summary.py $out/pmc3 >> $out/pmc_summary.txt 2>&1
find $out -name "*.csv" -size +2000k -delete
tail -3 $out/pmc.log | cut -c1-200
head -12 $out/kernel_stats.csv | cut -c1-110,280-400
grep -i "conv" $out/pmc_summary.txt | head -20 | cut -c1-160
