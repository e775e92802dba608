#!/usr/bin/env python3
"""Control-flow skeleton of a loop of a kernel in `hipcc -S` output: per basic block the number of instructions and every branch / exec-mask instruction.
usage: loop_skeleton.py kernels.s 'trace_kernel<1, false, false, false, 0>' .LBB39_266 [blocks]"""
import re, subprocess, sys
path, want, hdr = sys.argv[1], sys.argv[2], sys.argv[3]
nblocks = int(sys.argv[4]) if len(sys.argv) > 4 else 24
lines = open(path).read().split("\n")
starts = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
fn = None
for k, (i, n) in enumerate(starts):
    if want in subprocess.run(["c++filt", n], capture_output=True, text=True).stdout:
        fn = lines[i:(starts[k + 1][0] if k + 1 < len(starts) else len(lines))]
        break
assert fn, "kernel not found"
on = False; cnt = 0; n = 0; nv = 0
for l in fn:
    if l.startswith(hdr + ":"):
        on = True
    if not on:
        continue
    m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", l)
    if m:
        if n: print(f"     ... {n} instr ({nv} valu)")
        n = nv = 0
        cnt += 1
        if cnt > nblocks: break
        print("== " + m.group(0))
    elif l.startswith("\t") and not l.strip().startswith((";", ".")):
        t = l.strip()
        if t.startswith(("s_cbranch", "s_and_saveexec", "s_or_saveexec", "s_xor_b64 exec", "s_or_b64 exec", "s_branch", "s_andn2_b64 exec")):
            if n: print(f"     ... {n} instr ({nv} valu)")
            n = nv = 0
            print("     " + t)
        else:
            n += 1; nv += t.startswith("v_")
