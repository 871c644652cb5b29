#!/bin/bash
# A second build of the library with ONE translation unit recompiled under extra flags, for same-box A/Bs through SAF_LIB_PATH:
#   bash tools/build_variant.sh <tag> <file.hip> <flags...>   ->  tools/bin/libsaf_<tag>.so   (run in the build container)
set -e
cd "$(dirname "$0")/../spatially_aware_ai_amd/csrc"
tag=$1; src=$2; shift 2
mkdir -p ../../tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" -c $src -o /tmp/variant_$tag.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libsaf_$tag.so $objs /tmp/variant_$tag.o
echo built tools/bin/libsaf_$tag.so
