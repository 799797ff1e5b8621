for f in 0 1; do for r in 1 2 4 7; do echo "== FILTER=$f ROLES=$r"; FILTER=$f ROLES=$r ROUNDS=2 python tools/sweep_fwd.py 2>&1 | grep "top_k=16 thr=0.0\|top_k=16 thr=0.9"; done; done
