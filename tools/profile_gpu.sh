#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of
# the bench workload.  Results land under gpurun_out/prof/<tag>/ ; copy the
# summaries you want judged into profiles/.
set -e -o pipefail
TAG=${1:-r01}
ARGS=${2:---steps 100 --warmup 10 --no-epoch --no-cpu-baseline --no-variants --no-finalize-ab}
PMC_ARGS=${3:---steps 20 --warmup 5 --no-epoch --no-cpu-baseline --no-variants --no-finalize-ab}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
# PMC_SETS="a b|c d" overrides the counter sets (one rocprofv3 pass per |-separated set), e.g.
# PMC_SETS="FETCH_SIZE|WRITE_SIZE" for the traffic of a long-running workload only
DEFAULT_SETS="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum|SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"
IFS='|' read -r -a SETS <<< "${PMC_SETS:-$DEFAULT_SETS}"
for pmc in "${SETS[@]}"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py $PMC_ARGS > $OUT/pmc_$name.log 2>&1 || echo "pmc pass $name failed" >> $OUT/errors.log
done
cd $REPO
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
