# Round 4's measurement script, tracked in round 5 as it was run then (profiles/r04_* name it).  Variant libraries (tools/bin/libhdiff_*.so:
# build products, not tracked) are built with tools/scripts/ab_build.sh today; knobs this script sets through the environment may have
# become compile-time -D switches of such a build since (tools/README.md).
# timing ablations of conv1x1_x3.hip (C1X3_ABL): builds tools/bin/libhdiff_c1abl<mask>.so with "build", times them otherwise
cd /root/repo 2>/dev/null || cd "${GRAFT_REPO_ROOT}"
if [ "$1" = build ]; then
  cd hybrid-diffusion-underwater-atmopheric-image-enhancement_amd/csrc
  for a in 1 2 4 8 16 31; do
    /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -ffp-contract=off -DC1X3_ABL=$a -c conv1x1_x3.hip -o /tmp/c1abl$a.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v conv1x1_x3.o) /tmp/c1abl$a.o -o ../../tools/bin/libhdiff_c1abl$a.so
  done
  exit 0
fi
for s in "16 128 384 256 1" "16 128 128 256 1"; do
echo -n "base $s : "; python3 tools/conv_once.py $s 2>&1 | grep conv
for a in 1 2 4 8 16 31; do echo -n "ABL=$a : "; HDIFF_LIB=$PWD/tools/bin/libhdiff_c1abl$a.so timeout -k 10 120 python3 tools/conv_once.py $s 2>&1 | grep conv; done
done
