for v in 10012 10112 11112 112; do echo "== variant $v"; MRS_DEV=1 MRS_SWEEP_MQ_VARIANT=$v python tools/quick_sweep_mq.py 2>/dev/null | grep -E "^ring  .*nq= (3|4|8)"; done
