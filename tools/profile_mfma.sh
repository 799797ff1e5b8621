#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + MFMA counters of the matrix-core kernels.
set -e -o pipefail
TAG=${1:-r02_mfma}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/tools/profile_mfma.py > $OUT/wall.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/profile_mfma.py > $OUT/trace.log 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters_available.txt || true
# (round 3 moved the products to the bf16 matrix cores: the BF16 op counter beside the F32 one)
for pmc in "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  REPS=2 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/tools/profile_mfma.py > $OUT/pmc_$name.log 2>&1 || echo "pmc pass $name failed" >> $OUT/errors.log
done
cd $REPO
python3 tools/summarize_prof.py $OUT mfma > $OUT/summary.txt 2>&1 || true
python3 tools/mfma_util.py $OUT >> $OUT/summary.txt 2>&1 || true
cat $OUT/wall.log $OUT/summary.txt
