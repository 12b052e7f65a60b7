#!/bin/bash
# Round 6: does zeroing the fourth score term (k1 q1, 2^-24 of a score) change the d_head 16 forward's time / clock / power?  Same MFMAs, fewer toggling multipliers.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 120 python3 tools/kernel_power.py "$@" 2>&1 | grep -v amdgpu.ids; }
echo "== base"; run fwd16 16 4
echo "== t4 (fourth term zeroed)"; HDIFF_LIB=$PWD/tools/bin/libhdiff_t4.so run fwd16 16 4
echo "== base"; run fwd16 16 4
echo "== bwd base"; run bwd16 4 3
echo "== bwd ktreg"; HDIFF_LIB=$PWD/tools/bin/libhdiff_ktreg.so run bwd16 4 3
echo "== bwd base"; run bwd16 4 3
