#!/usr/bin/env python3
"""Instruction census of a trace kernel's step loop (the deepest loop that holds the Cash-Karp / Euler march) from `hipcc -S` output.
usage: loop_isa.py kernels.s 'trace_kernel<1, false, false, true, 0>' [--dump]
Counts the instructions of the basic blocks on the PLAIN march (the blocks a step executes when no rare path is taken): a block
belongs to it when it is in the innermost loop and not reached only through a branch that the plain march does not take - approximated
by: all blocks of the innermost loop up to the first block that starts the rare path (the block after the `s_xor_b64 exec` of the
combined rare-path test) plus the loop latch blocks.  With --dump prints those blocks."""
import re, sys, subprocess, collections

def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()

def main():
    path, want = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    starts = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    fn = None
    for k, (i, n) in enumerate(starts):
        if want in demangle(n):
            end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
            fn = lines[i:end]
            break
    assert fn, "kernel not found"
    # blocks
    blocks, cur, name = [], [], "entry"
    depth = {}
    for l in fn:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        m2 = re.match(r"^; %bb\.(\d+):\s*(;.*)?$", l)
        if m or m2:
            blocks.append((name, cur)); cur = []
            name = m.group(1) if m else "bb." + m2.group(1)
            c = (m.group(2) if m else m2.group(2)) or ""
            d = re.search(r"Depth=(\d+)", c)
            depth[name] = int(d.group(1)) if d else None
            continue
        cur.append(l)
    blocks.append((name, cur))
    maxd = max(d for d in depth.values() if d)
    inner = [(n, b) for n, b in blocks if depth.get(n) == maxd]
    # the inner loop may continue in blocks whose comment lacks Depth (continuations) - keep it simple
    tot = collections.Counter()
    per = []
    for n, b in inner:
        ins = [x.strip().split()[0] for x in b if x.startswith("\t") and not x.strip().startswith((";", "."))]
        per.append((n, len(ins), sum(1 for x in ins if x.startswith("v_")), sum(1 for x in ins if x.startswith("v_mov")), ins))
    print(f"{len(inner)} blocks at depth {maxd}")
    for n, k, v, mv, ins in per:
        print(f"  {n:12s} instr {k:4d} valu {v:4d} v_mov {mv:3d}")
        if dump:
            for n2, b in inner:
                if n2 == n:
                    print("\n".join(b))
    # resource usage
    for l in fn:
        if re.search(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size)|; (NumVgprs|NumSgprs|ScratchSize|Occupancy)", l):
            print(l.strip())

main()
