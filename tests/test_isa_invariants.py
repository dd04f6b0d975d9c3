"""Compile-time invariants of the kernels that wait for memory by HAND-COUNTED s_waitcnt (no GPU needed: hipcc -S).

The fused layer kernel k_block_x6 issues its fragment reads, its input-tile loads and its LDS-DMA pieces from inline asm or
builtins and waits for them with counted `lgkmcnt(N)` / `vmcnt(N)` statements.  Two things can break that silently when the
compiler (or the source) changes:
  * the compiler copies / parks a register whose inline-asm read is still pending (it does not know) -- found in round 4 as
    1e-4 errors of the d = 256 kernel; `tools/isa_pending_read_scan.py` walks the ISA for such uses;
  * the ORDER and NUMBER of vector memory operations between an issue point and its counted wait is not what the count
    assumes (vmcnt retires in order: "all but the newest N").
Both are checked on the ISA hipcc produces for gfx950 from the product source."""
import os
import re
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "influentialrs_amd", "csrc", "decoder.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def decoder_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = os.path.join(tmp_path_factory.mktemp("isa"), "decoder.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(REPO, "include"),
                        SRC, "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().split("\n")


def _kernel(lines, prefix):
    i = next(k for k, l in enumerate(lines) if l.startswith(prefix) and ":" in l.split()[0])
    j = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
    return lines[i:j]


def test_no_use_of_pending_inline_asm_reads(decoder_isa):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import isa_pending_read_scan as scan
    names = sorted({l.split(":")[0] for l in decoder_isa if l.startswith("_Z10k_block_x6") and ":" in l.split()[0]})
    assert len(names) >= 12, names  # every instantiation of the fused layer kernel
    for n in names:
        assert scan.scan(scan.kernel_lines(decoder_isa, n)) == 0, n


def test_counted_vmcnt_waits_see_the_operations_they_count(decoder_isa):
    """k_block_x6 at d = 128 on float16 planes (RESID_LATE): 12 parameter loads and ONE wait, 12 LDS-DMA pieces (three steps),
    16 attention-tile loads, the start wait vmcnt(24); per step a tile wait (12, 12, 28, 28) and the publish wait (20) in
    front of each barrier, 4 DMA pieces behind it, the 16 residual loads behind the second step's barrier; vmcnt(8) in front
    of LayerNorm 1; then one publish wait (4) and 4 pieces per step."""
    for qp0 in (0, 1):
        body = _kernel(decoder_isa, f"_Z10k_block_x6ILi{qp0}ELi4ELb0ELi4ELi2ELb0EEv11BlockX6Args")
        seq = []
        for l in body:
            t = l.strip()
            op = t.split()[0] if t else ""
            if op.startswith("global_load_lds"):
                seq.append("D")
            elif op.startswith(("global_load", "buffer_load", "scratch_load")):
                seq.append("L")
            elif op.startswith(("global_store", "scratch_store")):
                seq.append("S")
            elif op.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("W%s," % re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif op == "s_barrier":
                seq.append("|")
        s = "".join(seq)
        head = ("L" * 12 + "W0," + "D" * 12 + "L" * 16 + "W24,|" + "W12,W20,|DDDD" + "W12,W20,|DDDD" + "L" * 16 + "W28,W20,|DDDD" + "W28,W20,|DDDD" + "W8,")
        assert s.startswith(head), (qp0, s[:len(head) + 40])
        rest = s[len(head):]
        steps = 32 - 4 * qp0 - 4  # executed steps behind the out-projection
        m = re.match(r"^((?:W4,\|DDDD)+)", rest)
        assert m, rest[:80]
        assert "scratch" not in "".join(body)  # no spills: a reload would be a vmem operation the counts do not know
        # stores appear only in the q | k | v tail (and the x' store of LayerNorm 3), never between a DMA issue and the wait that counts it short
        assert rest.count("W0,") <= 1 + 1, rest.count("W0,")  # the last publish (nothing in flight behind it)
        assert rest.count("|") == steps - 1, (rest.count("|"), steps)


@pytest.fixture(scope="module")
def score_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = os.path.join(tmp_path_factory.mktemp("isa_score"), "score.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(REPO, "include"),
                        os.path.join(REPO, "influentialrs_amd", "csrc", "score.hip"), "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().split("\n")


def test_no_matrix_result_is_read_early_behind_a_taken_branch(decoder_isa, score_isa):
    """gfx950 does not interlock a vector read of an MFMA destination still in flight; the compiler pads the distance with s_nop
    along the paths it accounts for.  Round 5 met a taken edge of a wave-uniform branch it had not accounted for (a lab form of the
    sequence-resident attention: the row maximum read stale registers, NaN rows): the attention kernels now wait explicitly
    (SEQ_MFMA_LANDED / ATTN_MFMA_LANDED), and tools/isa_mfma_branch_scan.py walks every kernel with matrix instructions of both
    kernel files -- every branch target within the hazard window behind every MFMA -- for such a read.  (The scan flags that lab
    form's 12 reads; here it must find none.)"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import isa_mfma_branch_scan as scan
    for lines, min_kernels in ((decoder_isa, 40), (score_isa, 60)):
        ks = [(n, ins) for n, ins in scan.parse(lines) if any(op and op.startswith("v_mfma") for _, op, _, _ in ins)]
        assert len(ks) >= min_kernels, len(ks)
        for n, ins in ks:
            assert scan.scan_kernel(n, ins) == 0, n
