#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for cfg in "plain" "sep" "cols" "sep_rccl" "cols_rccl"; do
  unset HITL_FORCE_DIST HITL_DIST_COLUMNS HITL_FORCE_RCCL
  case $cfg in
    sep) export HITL_FORCE_DIST=1;;
    cols) export HITL_FORCE_DIST=1 HITL_DIST_COLUMNS=1;;
    sep_rccl) export HITL_FORCE_DIST=1 HITL_FORCE_RCCL=1;;
    cols_rccl) export HITL_FORCE_DIST=1 HITL_DIST_COLUMNS=1 HITL_FORCE_RCCL=1;;
  esac
  echo "== $cfg: $(REPS=6 python3 tools/dev/solve_c4.py 2>&1 | grep "solve ms")"
done
