#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in ${MODES:-0 1 2 3 0}; do timeout -k 10 60 python3 tools/power_watch.py tools/bin/mfma_k16_probe $m 3 2>&1 | grep -v amdgpu.ids; done
