# Round 4's measurement script, tracked in round 5 as it was run then (profiles/r04_* name it).  Variant libraries (tools/bin/libhdiff_*.so:
# build products, not tracked) are built with tools/scripts/ab_build.sh today; knobs this script sets through the environment may have
# become compile-time -D switches of such a build since (tools/README.md).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "base:"; timeout -k 10 120 python3 tools/attn_once.py 16 2>&1 | grep fwd
for a in 1 2 4 8 16 31; do echo "ABL=$a:"; HDIFF_LIB=$PWD/tools/bin/libhdiff_abl$a.so timeout -k 10 120 python3 tools/attn_once.py 16 2>&1 | grep fwd; done
echo "base again:"; timeout -k 10 120 python3 tools/attn_once.py 16 2>&1 | grep fwd
