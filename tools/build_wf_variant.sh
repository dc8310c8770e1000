#!/bin/bash
# a variant of the hot-path library with another csrc/winograd_fused.hip:  tools/build_wf_variant.sh NAME SOURCE [-D...]
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift || true
C=swapping_autoencoder_pytorch_amd/csrc
others=$(ls $C/*.hip | grep -v winograd_fused.hip)
mkdir -p tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -I include -I $C "$@" $src $others -o tools/variants/$name.so
echo built tools/variants/$name.so
