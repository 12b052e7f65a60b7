#!/bin/bash
# Round 6: the d_head 16 forward's timing ablations (attention_h2.hip -DH2_ABL=<bit>: 1 no workgroup barrier, 2 no rolling K / Q reloads, 4 no global -> LDS
# staging, 8 no reference check, 16 no V reload) WITH clock and board power (tools/kernel_power.py): what carries the energy at the power limit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 120 python3 tools/kernel_power.py "$@" 2>&1 | grep -v amdgpu.ids; }
echo "== base"; run fwd16 16 3
for v in fa1 fa2 fa4 fa8 fa16; do echo "== $v"; HDIFF_LIB=$PWD/tools/bin/libhdiff_$v.so run fwd16 16 2; done
echo "== base"; run fwd16 16 3
