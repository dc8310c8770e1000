#!/bin/bash
# like build_variant.sh but for changes outside conv2d.hip: builds the current csrc/ tree (or a copy given as $2) as tools/variants/$1.so
set -e
cd "$(dirname "$0")/.."
name=$1; C=${2:-swapping_autoencoder_pytorch_amd/csrc}; shift; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DSAE_TUNING -I include -I $C "$@" $C/*.hip -o tools/variants/$name.so
echo built tools/variants/$name.so
