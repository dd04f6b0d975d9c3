#!/usr/bin/env python3
"""Scan a kernel's ISA for uses of registers whose inline-asm LDS read (or global load) has not been waited for.

The fused layer kernels issue their fragment reads as inline-asm `ds_read_b128` and wait for them with hand-counted
`s_waitcnt lgkmcnt(N)` statements (csrc/decoder.hip: k_block_x6, csrc/score.hip: the ring sweeps).  The compiler does not
know the destination is pending: under register pressure it may copy such a register (a move into an AGPR, a live-range
split) between the read and its wait, and the copy holds whatever the register held before.  This tool walks the
instruction stream, keeps the destinations of reads that no `lgkmcnt` wait has retired yet, and prints every instruction
that touches one.  Round 4: it found the cause of 1e-4 errors in the d = 256 layer kernel (fragments read ahead across a
LayerNorm phase were parked in AGPRs); the d = 128 kernels scan clean.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only influentialrs_amd/csrc/decoder.hip -o dec.s
       tools/isa_pending_read_scan.py dec.s <mangled-kernel-name-prefix> [...]"""
import re
import sys


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add(m.group(1) + m.group(2))
    return out


def kernel_lines(lines, prefix):
    on, out = False, []
    for l in lines:
        if l.startswith(prefix) and l.rstrip().endswith(":") or (l.startswith(prefix) and ":" in l.split()[0]):
            on = True
        if on:
            out.append(l)
            if "s_endpgm" in l:
                break
    return out


VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_load",
        "scratch_store", "flat_load", "flat_store", "flat_atomic")


def scan(lines, limit=40):
    """pending: LDS reads not yet retired by an lgkmcnt wait; vpending: vector memory operations (in issue order, every
    kind: they share vmcnt and retire in order) not yet retired by a vmcnt wait -- an entry carries the destination
    registers of a load into registers, or an empty set (stores, LDS-DMA loads).  Round 4: the layer kernel requests its
    attention tiles by inline-asm global loads with hand-counted vmcnt waits, the same hazard class as the fragment reads."""
    pending, vpending, flags, in_asm = [], [], 0, False
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        if not t or t[0] in ";.":
            continue
        parts = t.split(None, 1)
        op, ops = parts[0], parts[1] if len(parts) > 1 else ""
        if op.startswith("ds_read") or op.startswith("ds_load"):
            # (only inline-asm reads carry destinations: the compiler waits for its own reads, and a linear walk over a kernel
            #  with branches would pair a read with the other arm's writes)
            pending.append((regs_of(ops.split(",")[0]) if in_asm else set(), i, t))
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n > 0 else []
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                vpending = vpending[len(vpending) - n:] if n > 0 else []
            continue
        used = regs_of(ops)
        for r, ln, tt in pending + vpending:
            if used & r:
                flags += 1
                if flags <= limit:
                    print(f"  line {i}: {t}    <-- register of the pending read at line {ln}: {tt}")
        if op.startswith(VMEM):
            is_load = "_load" in op and "_lds_" not in op and " lds" not in t
            vpending.append((regs_of(ops.split(",")[0]) if is_load and in_asm else set(), i, t))
    return flags


def main():
    lines = open(sys.argv[1]).read().split("\n")
    bad = 0
    for prefix in sys.argv[2:]:
        k = kernel_lines(lines, prefix)
        n = scan(k)
        print(f"{prefix}: {len(k)} lines, {n} uses of pending read destinations")
        bad += n
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
