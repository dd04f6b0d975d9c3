#!/usr/bin/env python3
"""Scan kernels' ISA for a matrix-instruction result that is read too soon behind a TAKEN branch.

gfx950 does not interlock a vector instruction that reads the destination of an MFMA still in flight; the compiler pads the
distance with s_nop along the path it lays out, and round 5 found a case where the TAKEN edge of a wave-uniform branch right
behind an MFMA reached a reader after 2 instructions (a lab form of the sequence-resident attention: NaN rows).  This tool walks
every kernel of an ISA listing: for each MFMA it follows the instruction stream AND every branch target for `WINDOW` issue slots
(an s_nop N counts N + 1) and reports any non-MFMA instruction inside the window that reads a destination register of that MFMA
on a path that crossed a taken branch.  Fall-through-only paths are the compiler's business (it pads those); a report is a
place where the source needs an explicit wait (SEQ_MFMA_LANDED / ATTN_MFMA_LANDED in csrc/decoder.hip).

The window is the number of wait states the instruction needs before a vector read (passes + 4 on gfx950: 8 behind a 4-pass
16x16x32 f16 / bf16, 12 behind an 8-pass 32x32x16 f16 or 16x16x4 f32, 20 behind 16 passes); an independent MFMA in between counts
as its own passes (the matrix pipe is busy that long), every other instruction as one, s_nop N as N + 1.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only <file>.hip -o k.s ; tools/isa_mfma_branch_scan.py k.s"""
import re
import sys


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add(m.group(1) + m.group(2))
    return out


def passes_of(op):
    """matrix-pipe passes (4 cycles each) of an MFMA from its shape and type: flops / (flops per cycle and SIMD) / 4"""
    m = re.search(r"_(\d+)x(\d+)x(\d+)", op)
    if not m:
        return 16
    a, b, k = (int(x) for x in m.groups())
    if re.search(r"(f16|bf16)$", op):
        rate = 1024
    elif re.search(r"(i8|fp8|bf8|f8)", op):
        rate = 2048
    elif re.search(r"x\d+_?f64$", op):
        rate = 32
    else:
        rate = 64  # float32 (and xf32-less gfx950)
    return max(1, a * b * k * 2 // rate // 4)


def need_states(op):
    # the compiler's own figure on gfx950 for "XDL write VGPR -> VALU read": passes + 3 + 1 (observed: 8 behind a 4-pass MFMA)
    return passes_of(op) + 4


def parse(lines):
    """-> list of kernels: (name, [(label or None, op, operands, raw)])"""
    kernels, cur, name = [], None, None
    for l in lines:
        if re.match(r"^[A-Za-z_][\w$.]*:\s*(;.*)?$", l) and not l.startswith(".L"):
            name = l.split(":")[0]
            cur = []
            kernels.append((name, cur))
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur.append((m.group(1), None, None, l))
            continue
        t = l.strip()
        if not t or t[0] in ";." or not l.startswith("\t"):
            continue
        parts = t.split(None, 1)
        cur.append((None, parts[0], parts[1] if len(parts) > 1 else "", t))
        if parts[0] == "s_endpgm":
            cur = None
    return kernels


def scan_kernel(name, ins, limit=20):
    labels = {lab: i for i, (lab, op, _, _) in enumerate(ins) if lab}
    found = 0
    for i, (lab, op, ops, raw) in enumerate(ins):
        if not op or not op.startswith("v_mfma"):
            continue
        dst = regs_of(ops.split(",")[0])
        window = need_states(op)
        # depth-first over (position, issue slots used, crossed a taken branch)
        stack, seen = [(i + 1, 0, False)], set()
        while stack:
            pos, used, taken = stack.pop()
            while pos < len(ins) and used < window:
                key = (pos, taken)
                if key in seen:
                    break
                seen.add(key)
                lab2, op2, ops2, raw2 = ins[pos]
                if op2 is None:
                    pos += 1
                    continue
                if op2 == "s_nop":
                    used += int(ops2.strip() or 0) + 1
                    pos += 1
                    continue
                if op2 == "s_endpgm":
                    break
                if op2.startswith("s_cbranch") or op2 == "s_branch":
                    tgt = ops2.strip()
                    if tgt in labels:
                        stack.append((labels[tgt], used + 1, True))
                    if op2 == "s_branch":
                        break
                    used += 1
                    pos += 1
                    continue
                srcs = regs_of(ops2.split(",", 1)[1]) if "," in ops2 else set()
                if op2.startswith("v_mfma"):
                    # a following MFMA reading it as SrcC / overwriting it has its own (shorter, compiler-handled) rules; an
                    # independent one keeps the matrix pipe busy for its own passes before anything behind it issues
                    if regs_of(ops2.split(",")[0]) & dst:
                        break
                    used += passes_of(op2)
                    pos += 1
                    continue
                elif (srcs & dst) and taken:
                    if op2.startswith("v_accvgpr_read") and used >= window - 2:
                        break  # (AGPR read-back: the compiler's figure is passes + 2)
                    found += 1
                    if found <= limit:
                        print(f"  {name}: `{raw}` (needs {window} wait states) is read after {used} behind a taken branch by `{raw2}`")
                    break
                elif regs_of(ops2.split(",")[0]) & dst and not op2.startswith(("global_store", "ds_write", "scratch_store", "buffer_store")):
                    break  # overwritten
                used += 1
                pos += 1
    return found


def main():
    lines = open(sys.argv[1]).read().split("\n")
    total = 0
    nk = 0
    for name, ins in parse(lines):
        if len(sys.argv) > 2 and not any(name.startswith(p) for p in sys.argv[2:]):
            continue
        if not any(op and op.startswith("v_mfma") for _, op, _, _ in ins):
            continue
        nk += 1
        total += scan_kernel(name, ins)
    print(f"{nk} kernels with matrix instructions scanned: {total} early reads behind taken branches")
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
