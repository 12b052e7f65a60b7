#!/bin/bash
# Dev tool (runs in the build container, no GPU): build a VARIANT of one csrc/*.hip file into a second shared library,
#   bash tools/scripts/ab_build.sh <name> <file stem> <extra hipcc flags...>      -> tools/bin/libhdiff_<name>.so
# e.g. bash tools/scripts/ab_build.sh h2w_abl8 attention_h2w -DH2W_ABL=8
# The variant is selected on the GPU box with HDIFF_LIB=$PWD/tools/bin/libhdiff_<name>.so (hdiff_amd/_capi.py), so that both
# builds are timed inside ONE gpurun call on the same box (tools/scripts/ab_run.sh).  tools/bin/ holds only build products.
set -e
cd "$(dirname "$0")/../.."
NAME=$1; STEM=$2; shift 2
C=hybrid-diffusion-underwater-atmopheric-image-enhancement_amd/csrc
make -s -C $C >/dev/null
FLAGS=$(make -s -C $C --eval "pf: ; @echo \$(CXXFLAGS) \$(FLAGS_$STEM)" pf)
mkdir -p tools/bin/obj
/opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$STEM.hip -o tools/bin/obj/${STEM}_$NAME.o
OBJS=$(ls $C/build/*.o | grep -v "/$STEM.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS tools/bin/obj/${STEM}_$NAME.o -o tools/bin/libhdiff_$NAME.so
echo "built tools/bin/libhdiff_$NAME.so ($STEM $*)"
