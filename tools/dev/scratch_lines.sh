#!/bin/bash
# dev probe: scratch accesses of landing_ipm_kernel by source line (device-only build with line tables)   tools/dev/scratch_lines.sh [-DFLAG ...]
cd "$(dirname "$0")/../../landing-controller_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -enable-ipra=0 -mllvm -pragma-unroll-threshold=1000000 --cuda-device-only -gline-tables-only -c capi.hip -o /tmp/devb/lines.o "$@" 2>/dev/null || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/devb/lines.o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=/tmp/devb/lines.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d -l --no-show-raw-insn /tmp/devb/lines.co > /tmp/devb/lines.s
python3 - <<'PY'
import re,collections
cur=None; line=None; cnt=collections.Counter()
for l in open('/tmp/devb/lines.s'):
    m=re.match(r'^[0-9a-f]+ <(.*)>:',l)
    if m: cur=m.group(1); continue
    m=re.match(r'^; (.*):(\d+)$',l.strip())
    if m: line=(m.group(1).split('/')[-1],int(m.group(2))); continue
    if cur and 'landing_ipm_kernel' in cur and ('scratch_load' in l or 'scratch_store' in l):
        cnt[(line,'ld' if 'load' in l else 'st')]+=1
agg=collections.Counter()
for (ln,k),v in cnt.items(): agg[ln]+=v
out=[(ln,v,cnt[(ln,'ld')],cnt[(ln,'st')]) for ln,v in agg.items()]
# group by file and 25-line buckets
b=collections.Counter()
for ln,v,_,_ in out: b[(ln[0], ln[1]//25*25)]+=v
for k,v in sorted(b.items()): print("%-24s %5d..%-5d %5d" % (k[0],k[1],k[1]+24,v))
PY
