# Round 4's measurement script, tracked in round 5 as it was run then (profiles/r04_* name it).  Variant libraries (tools/bin/libhdiff_*.so:
# build products, not tracked) are built with tools/scripts/ab_build.sh today; knobs this script sets through the environment may have
# become compile-time -D switches of such a build since (tools/README.md).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo -n "base : "; python3 tools/conv_once.py 16 128 128 256 3 pairs 2>&1 | grep conv
for a in 1 2 4 8 16 17 31; do echo -n "ABL=$a : "; HDIFF_LIB=$PWD/tools/bin/libhdiff_cabl$a.so timeout -k 10 120 python3 tools/conv_once.py 16 128 128 256 3 pairs 2>&1 | grep conv; done
echo -n "base : "; python3 tools/conv_once.py 16 128 128 256 3 pairs 2>&1 | grep conv
