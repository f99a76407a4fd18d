#!/bin/bash
# dev probe: device-only compile of the product sources with extra flags -> kernel resource usage + static scratch counts
#   tools/dev/devbuild.sh NAME [-DFLAG ...]        (output under /tmp/devb/NAME.*)
name=$1; shift
mkdir -p /tmp/devb
cd "$(dirname "$0")/../../landing-controller_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -enable-ipra=0 -mllvm -pragma-unroll-threshold=1000000 --cuda-device-only -c capi.hip -o /tmp/devb/$name.o -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/devb/$name.rem || { tail -20 /tmp/devb/$name.rem; exit 1; }
grep -A12 "landing_ipm_kernel" /tmp/devb/$name.rem | grep -E "VGPRs:|AGPRs|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | tr '\n' ' '; echo
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/devb/$name.o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=/tmp/devb/$name.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/devb/$name.co > /tmp/devb/$name.s
python3 /root/repo/tools/dev/scratch_count.py /tmp/devb/$name.s | grep -E "ipm_kernel|eval_task|member_eval|riccati|condense|forward_pass|row_products|block_elim|gauss" | sed 's/_ZN7landing[0-9]*//; s/E[A-Z].*//' 
