for a in 1 2 3 6; do echo "== ahead $a"; HITL_LM_AHEAD=$a HITL_TIMING=1 python tools/dev/lm_ab.py 2>&1 | grep "lm_device_loop\|host_lm" | tail -4 | head -3; done
