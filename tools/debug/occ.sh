set -e
cd sngnn_amd/csrc
for cfg in "-DSNGNN_FWD_WAVES=5 -DSNGNN_FILT_UF=8 -DSNGNN_LIST_U=2" "-DSNGNN_FWD_WAVES=5" "-DSNGNN_FWD_WAVES=4 -DSNGNN_FILT_UF=8 -DSNGNN_LIST_U=2"; do
  make -j16 EXTRA="$cfg" > /dev/null 2>&1
  cd ../..
  echo "== $cfg"
  FILTER=1 ROUNDS=2 python tools/sweep_fwd.py 2>&1 | grep "top_k=16"
  cd sngnn_amd/csrc
done
make -j16 > /dev/null 2>&1
