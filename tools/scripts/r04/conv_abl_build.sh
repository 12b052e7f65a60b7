# Round 4's measurement script, tracked in round 5 as it was run then (profiles/r04_* name it).  Variant libraries (tools/bin/libhdiff_*.so:
# build products, not tracked) are built with tools/scripts/ab_build.sh today; knobs this script sets through the environment may have
# become compile-time -D switches of such a build since (tools/README.md).
# builds tools/bin/libhdiff_cabl<mask>.so: the library with the fp16-pair 3x3 kernel's timing ablation <mask> (CONVH2_ABL)
cd /root/repo/hybrid-diffusion-underwater-atmopheric-image-enhancement_amd/csrc
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -DCONVH2_ABL=$a -c conv3x3_x3.hip -o /tmp/conv_abl$a.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v conv3x3_x3.o) /tmp/conv_abl$a.o -o /root/repo/tools/bin/libhdiff_cabl$a.so
done
