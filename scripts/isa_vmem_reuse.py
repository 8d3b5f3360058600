#!/usr/bin/env python
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for vector-memory instructions whose ADDRESS or store-DATA registers are overwritten by one of the next few
instructions: the write then waits until the memory pipeline has read the operand (measured in k_pwb, round 6: ~150 cycles per store under load).
    python scripts/isa_vmem_reuse.py file.s [window]   -> per kernel: memory instructions, how many are followed by such a write"""
import re
import sys

win = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rx_mem = re.compile(r"^\s*(global_load|global_store|buffer_load|buffer_store|flat_load|flat_store)\w*\s+(.*)")
rx_reg = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    out = set()
    for m in rx_reg.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


kernel, stats, lines = None, {}, []
for ln in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kernel = m.group(1); lines = []
        stats[kernel] = [0, 0]
        continue
    if kernel is None or not ln.startswith("\t") or ln.lstrip().startswith((".", ";")):
        continue
    lines.append(ln.strip())
    if len(lines) > win + 1:
        lines.pop(0)
    # examine the instruction `win` back: is one of its source registers written by a later one in the window?
    first = lines[0]
    mm = rx_mem.match("\t" + first)
    if mm and len(lines) == win + 1:
        ops = [o.strip() for o in mm.group(2).split(",")]
        is_store = "store" in mm.group(1)
        src = set()
        for o in (ops if is_store else ops[1:]):          # loads: every operand but the destination
            src |= regs(o)
        hit = False
        for later in lines[1:]:
            parts = later.split(None, 1)
            if len(parts) < 2 or parts[0].startswith(("s_", "global_store", "buffer_store", "ds_write")):
                continue
            dst = regs(parts[1].split(",")[0])
            if dst & src:
                hit = True
        stats[kernel][0] += 1
        stats[kernel][1] += int(hit)
for k, (n, h) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if h:
        print("%6d of %6d  %s" % (h, n, k))
