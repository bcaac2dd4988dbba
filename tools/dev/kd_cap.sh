for cap in 0 7 5 3; do for blk in 256 512; do HITL_KD_CAP=$cap HITL_KD_BLOCK=$blk HITL_KD_REFILL=16 python tools/kd_time.py 2>&1 | grep "^kd block" | sed "s/^/cap $cap: /"; done; done
