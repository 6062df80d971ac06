#!/bin/bash
# build_ab/<name>/gpt_image_edit_amd/libfk_gfx950.so = the current objects with ONE source recompiled under extra flags:
#   tools/build_variant.sh <name> <source.hip> [extra hipcc flags ...]      (A/B builds for FK_LIB_PATH; build_ab/ is git-ignored)
set -e
cd "$(dirname "$0")/../gpt_image_edit_amd/csrc"
name=$1; src=$2; shift 2
make -s
out=../../build_ab/$name/gpt_image_edit_amd
mkdir -p $out /tmp/fk_variant_$name
extra=""
{ [ "$src" = attention_bwd.hip ] || [ "$src" = attention_fwd4.hip ]; } && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $extra "$@" -c $src -o /tmp/fk_variant_$name/${src%.hip}.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fk_variant_$name/${src%.hip}.o -o $out/libfk_gfx950.so
echo "built $out/libfk_gfx950.so"
