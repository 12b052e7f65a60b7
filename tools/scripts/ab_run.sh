#!/bin/bash
# Dev tool (GPU box, through gpurun): time one driver under the product library and under variant libraries, same box, same call.
#   bash tools/scripts/ab_run.sh "<python driver and args>" <variant name>...     e.g.  "tools/attn_once.py 16" h2w_abl1 h2w_abl2
# Variants are tools/bin/libhdiff_<name>.so (tools/scripts/ab_build.sh).  Every run is bounded by its own timeout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
DRV=$1; shift
run() { timeout -k 10 ${AB_TIMEOUT:-180} python3 $DRV 2>&1 | grep -v amdgpu.ids; }
echo "== base"; run
for v in "$@"; do echo "== $v"; HDIFF_LIB=$PWD/tools/bin/libhdiff_$v.so run; done
echo "== base again"; run
