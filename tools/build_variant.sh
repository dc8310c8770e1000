#!/bin/bash
# build a variant of the hot-path library for A/B timing:  tools/build_variant.sh NAME [conv2d source] [-D...]
# (tools/variants/*.so are git-ignored but travel to the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; src=${2:-swapping_autoencoder_pytorch_amd/csrc/conv2d.hip}; shift; shift || true
C=swapping_autoencoder_pytorch_amd/csrc
others=$(ls $C/*.hip | grep -v conv2d.hip)
mkdir -p tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DSAE_TUNING -I include -I $C "$@" $src $others -o tools/variants/$name.so
echo built tools/variants/$name.so
