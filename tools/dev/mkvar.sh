#!/bin/bash
# dev probe: build a variant of the product library with extra flags -> landing-controller_amd/_var/lib_NAME.so (ships with gpurun, git-ignored)
#   tools/dev/mkvar.sh NAME [-DFLAG ...]
name=$1; shift
root="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p $root/landing-controller_amd/_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -enable-ipra=0 -mllvm -pragma-unroll-threshold=1000000 -shared "$@" -o $root/landing-controller_amd/_var/lib_$name.so $root/landing-controller_amd/csrc/capi.hip 2>&1 | grep -E "error" -A3 | head -20
ls -la $root/landing-controller_amd/_var/lib_$name.so | awk '{print $5, $9}'
