for r in 1 2 7; do echo "== FILTER=1 ROLES=$r"; FILTER=1 ROLES=$r ROUNDS=2 python tools/sweep_fwd.py 2>&1 | grep "top_k=16 thr=0.0\|top_k=16 thr=0.9"; done
