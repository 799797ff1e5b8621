#!/bin/bash
# Run on the GPU box (through gpurun): every measurement tool once, outputs concatenated under
# gpurun_out/<tag>_tools_record.txt (copy into profiles/).
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_tools_record.txt
: > $OUT
for t in bench_modes.py bench_configs.py bench_epochs.py sweep_fwd.py bench_bwd.py bench_cosine_cmp.py bench_knn.py bench_plumbing.py lin_error.py bench_pp.py; do
  echo "== tools/$t ==" >> $OUT
  python tools/$t 2>&1 | grep -v amdgpu.ids >> $OUT
  echo >> $OUT
done
echo "== FILTER=2 tools/sweep_fwd.py (fp16 filter forced on) ==" >> $OUT
FILTER=2 python tools/sweep_fwd.py 2>&1 | grep -v amdgpu.ids >> $OUT
echo "== FILTER=0 tools/sweep_fwd.py (fp16 filter off) ==" >> $OUT
FILTER=0 python tools/sweep_fwd.py 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
