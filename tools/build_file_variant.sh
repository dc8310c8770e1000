#!/bin/bash
# a variant of the hot-path library with ONE source replaced, linked against the product's other objects (csrc/build/product/*.o):
#   tools/build_file_variant.sh NAME upfirdn2d.hip SOURCE [-D...]
set -e
cd "$(dirname "$0")/.."
name=$1; file=$2; src=$3; shift; shift; shift || true
C=swapping_autoencoder_pytorch_amd/csrc
others=$(ls $C/build/product/*.o | grep -v "/$file.o")
mkdir -p tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I include -I $C "$@" -c $src -o /tmp/var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name.o $others -o tools/variants/$name.so
echo built tools/variants/$name.so
