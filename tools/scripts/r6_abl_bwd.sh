#!/bin/bash
# Round 6: clock / power of the attention backward (pipelined vs two-barrier kernel) and the timing ablations of the pipelined kernel,
# one gpurun call, every run under its own timeout.  Variants: tools/scripts/ab_build.sh p<bits> attention_bwd_h2 -DH2P_ABL=<bits>, old = -DH2B_PIPE=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 120 python3 tools/kernel_power.py "$@" 2>&1 | grep -v amdgpu.ids; }
echo "== base"; run bwd16 4 4
echo "== old"; HDIFF_LIB=$PWD/tools/bin/libhdiff_old.so run bwd16 4 4
for v in p1 p2 p4 p6 p8 p16 p32 p64 p128; do echo "== $v"; HDIFF_LIB=$PWD/tools/bin/libhdiff_$v.so run bwd16 4 2; done
echo "== base again"; run bwd16 4 4
echo "== fwd16"; run fwd16 16 4
