for r in 1 2 4 7; do echo "== OTF ROLES=$r"; ROLES=$r ROUNDS=2 python tools/sweep_fwd.py 2>&1 | grep "top_k="; done
