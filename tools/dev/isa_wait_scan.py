#!/usr/bin/env python3
"""Lists, per kernel of a translation unit, the places where a vector load is followed by a FULL wait (s_waitcnt vmcnt(0)) and then
another load: dependent round trips, or loads the compiler serialised (a load under a divergent condition is closed with a wait; a
store loop whose stores may alias the next load keeps load, wait, store in sequence).  How select_compact_kernel's four serialised
key / state loads, the Schur tile kernel's five map loads and the mirror kernels' four tile loads were found (docs/LOG_r05.md).
usage: tools/dev/isa_wait_scan.py ptam_cg_amd/csrc/bundle.hip [kernel name filter ...]   (compiles the file to ISA with hipcc)"""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
filt = sys.argv[2:]
out = tempfile.mktemp(suffix=".s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Wno-unused-function",
                "-Wno-unused-value", "-S", "--cuda-device-only", "-o", out, os.path.basename(src)], cwd=os.path.dirname(os.path.abspath(src)),
               check=True, stderr=subprocess.DEVNULL)
t = open(out).read().split("\n")
os.unlink(out)
cur, base, ev = None, 0, []
def flush():
    if not cur or (filt and not any(f in cur for f in filt)):
        return
    hops = []
    for i, (ln, k) in enumerate(ev):
        if k == "W" and i > 0 and ev[i - 1][1] == "L" and ln - ev[i - 1][0] <= 12:
            nxt = [e for e in ev[i + 1:i + 4] if e[1] == "L"]
            if nxt and nxt[0][0] - ln <= 40:
                hops.append(ln)
    if hops:
        print("%-70s loads %4d, load -> full wait -> load at lines %s" % (cur[:70], sum(1 for e in ev if e[1] == "L"), hops[:16]))
for i, l in enumerate(t):
    m = re.match(r"^(_Z\w+):\s+; @", l)
    if m:
        flush()
        cur, base, ev = m.group(1), i, []
    if cur:
        if re.search(r"\b(global_load|flat_load|buffer_load)", l):
            ev.append((i - base, "L"))
        elif "s_waitcnt vmcnt(0)" in l:
            ev.append((i - base, "W"))
flush()
