"""dev probe: static count of scratch (private-segment) loads / stores per function of the gfx950 code object.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -enable-ipra=0 -mllvm -pragma-unroll-threshold=1000000 --cuda-device-only -c landing-controller_amd/csrc/capi.hip -o /tmp/capi_dev.o
   /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/capi_dev.o > /tmp/capi.s ;  python tools/dev/scratch_count.py /tmp/capi.s [filter]"""
import re, subprocess, sys
cur, cnt, tot = None, {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
        cur = m.group(1); cnt[cur] = [0, 0, 0]; continue
    if cur is None:
        continue
    s = line.strip()
    if not s:
        continue
    cnt[cur][2] += 1
    if "scratch_load" in s or ("buffer_load" in s and "offen" in s and "s[0:3]" in s):
        cnt[cur][0] += 1
    if "scratch_store" in s or ("buffer_store" in s and "s[0:3]" in s):
        cnt[cur][1] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = list(cnt)
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.split("\n")
except Exception:
    dem = names
for n, d in zip(names, dem):
    if flt in d and (cnt[n][0] + cnt[n][1] > 0 or flt):
        print("%6d ld %6d st %7d insts  %s" % (cnt[n][0], cnt[n][1], cnt[n][2], d[:150]))
