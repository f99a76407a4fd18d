#!/bin/bash
# dev: builds the product library with extra -D flags into tools/dev/variants/<tag>.so      tools/dev/build_variant.sh tag -DX=1 ...
tag=$1; shift
d=$(dirname $0)/variants; mkdir -p $d
cd $(dirname $0)/../../landing-controller_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -enable-ipra=0 "$@" -shared -o ../../tools/dev/variants/$tag.so capi.hip 2> /dev/null && echo built $tag
