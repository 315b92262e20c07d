"""CPU (hipcc cross-compiles without a GPU): properties of the generated gfx950 code that the kernels' performance and, for
the LDS-DMA rings, their design rest on -- checked on the assembly, because hipcc is free to change them silently.

* no instantiation of any kernel uses scratch (register spills);
* the K loop of every hgemm / hwgrad instantiation keeps its ring in flight: one counted `s_waitcnt vmcnt(N > 0)` before
  the barrier and no compiler-inserted `vmcnt(0)` (hipcc did insert one in hwgrad_kernel)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wavenet_speech_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


# per-file flags of csrc/Makefile (FLAGS_<file>): the assembly checked here must be the product's
EXTRA_FLAGS = {"wn_half.hip": ["-fno-slp-vectorize"], "wn_fused.hip": ["-fno-slp-vectorize"], "wn_col.hip": ["-fno-slp-vectorize"],
               "wn_col_conv.hip": ["-fno-slp-vectorize"], "wn_col2.hip": ["-fno-slp-vectorize"], "wn_col_skip.hip": ["-fno-slp-vectorize"]}


_ASM = {}


def _asm(src, tmp_path):
    """assembly listing of one kernel file, compiled once per test session (several tests read the same listing)"""
    if src in _ASM and os.path.exists(_ASM[src]):
        return _ASM[src]
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="wn_asm_"), src + ".s")
    _ASM[src] = out
    r = subprocess.run([HIPCC, "-O3", "-std=c++20", "--offload-arch=gfx950", "--cuda-device-only"] + EXTRA_FLAGS.get(src, []) +
                       ["-S", os.path.join(CSRC, src), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def test_ring_kernels_keep_their_prefetch_in_flight(tmp_path):
    files = [_asm("wn_half.hip", tmp_path), _asm("wn_half_wgrad.hip", tmp_path), _asm("wn_fused.hip", tmp_path)]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rings.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    # 15 hgemm (12 at 128 rows, 3 at 256) + 24 hgemm8 + 3 hwgrad + 24 fused-forward instantiations
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 66, r.stdout


@pytest.mark.parametrize("src", ["wn_gemm.hip", "wn_wgrad.hip", "wn_half.hip", "wn_half_wgrad.hip", "wn_fused.hip", "wn_col.hip", "wn_col_conv.hip", "wn_col_skip.hip", "wn_col2.hip", "wn_embed.hip",
                                 "wn_nll.hip", "wn_pack.hip"])
def test_no_kernel_uses_scratch(src, tmp_path):
    text = open(_asm(src, tmp_path)).read()
    sizes = re.findall(r"\.private_segment_fixed_size:\s*(\d+)", text)
    assert sizes, "no kernel metadata found"
    assert all(int(x) == 0 for x in sizes), "scratch in use: %s" % sizes
    assert "scratch_store" not in text and "scratch_load" not in text


def test_column_owner_kernels_wait_for_exactly_their_stage(tmp_path):
    """hcol_kernel (wn_col.hip) counts, per ring stage, everything that is younger than the stage's LDS-DMA pieces in the in-order
    vmcnt queue: the pieces of the stages after it and the compiler-visible loads its schedule has issued since.  The counts
    are template arithmetic; here the schedule is restated independently and compared with the `s_waitcnt vmcnt(N)` in front of
    every K-loop barrier of every instantiation (a count too high would read a stage before it has landed)."""
    KCD, KCRES, KCPW = 6, 16, 2

    def frag_at(j):
        return 0 if j < KCRES else 2 * ((j - KCRES) // 4 + 1)

    def epi_at(m, nks):
        x = (nks + 4 * (m + 1) - KCRES + 1) // 2 if nks + 4 * (m + 1) - KCRES + 1 >= 0 else 0
        return min(max(x, 0), nks // 2)

    def epi_loads(epi):                      # per row tile: dgate reads z and sigmoid (8), the leaky epilogues bias or mask (4)
        return {2: 8, 100: 4, 101: 4}.get(epi, 0)

    def visible_at(X, nks, nt, epi):
        n = sum(1 for j in range(KCRES, nks) if X > 0 and frag_at(j) == X)
        n += sum(epi_loads(epi) for m in range(nt) if epi_at(m, nks) == X)
        return n

    def expected(nks, nt, epi):
        nst = nks // 2
        out = []
        for S in range(nst):
            ahead = min(nst - 1 - S, KCD - 2)
            out.append(KCPW * ahead + sum(visible_at(X, nks, nt, epi) for X in range(max(0, S - KCD + 2), S + 1)))
        return out

    seen = 0
    for src in ("wn_col.hip", "wn_col_conv.hip", "wn_col_skip.hip"):
        text = open(_asm(src, tmp_path)).read()
        for m in re.finditer(r"^(_ZN2wn11hcol_kernel\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S | re.M):
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            bf, nt, nks, epi = re.search(r"hcol_kernel<(\w+), (\d+), (\d+), (\d+)>", name).groups()
            lines = [l.strip() for l in m.group(2).splitlines()]
            bars = [i for i, l in enumerate(lines) if l == "s_barrier"]
            waits = []
            for b in bars[1:]:                   # (the first barrier closes the prologue: its wait is the visible vmcnt(0))
                w = [l for l in lines[max(0, b - 3):b] if re.match(r"s_waitcnt vmcnt\(\d+\)$", l)]
                assert w, (name, "no counted wait in front of a K-loop barrier")
                waits.append(int(re.match(r"s_waitcnt vmcnt\((\d+)\)", w[-1]).group(1)))
            assert waits == expected(int(nks), int(nt), int(epi)), (name, waits, expected(int(nks), int(nt), int(epi)))
            assert len(bars) == int(nks) // 2 + 1, name
            seen += 1
    # (8 dz + 8 dx + 8 masked dx shapes) + (16 conv shapes x forward / backward) + (skips_sum of 1..16 blocks at 64 / 128 channels), x (bf16, f16)
    assert seen == 48 + 64 + 64, seen


def test_paired_dx_dz_kernel_waits_for_exactly_its_stage(tmp_path):
    """hcol2_kernel (wn_col2.hip: dx of a block and dz of the block below it in one launch): the same exact-count discipline, with
    two products on one stage stream, a fragment stream that continues from dx's operands into dz's dS segment, the dgate inputs
    scheduled as dS dies, and the dx STORES between the two products counted for the stages whose pieces were issued before them."""
    KCD, KCRES, KCPW = 6, 16, 2

    def frag_at(j):
        return 0 if j < KCRES else 2 * ((j - KCRES) // 4 + 1)

    def expected(nt, hasdr):
        nks1 = (10 if hasdr else 8) * nt
        nst1, nf, nst = nks1 // 2, nks1 + 2 * nt, nks1 // 2 + 2 * nt
        epi_at = [min(nst1 + 2 * (m + 1), nst) for m in range(nt)]

        def visible_at(X):
            return sum(1 for j in range(KCRES, nf) if X > 0 and frag_at(j) == X) + sum(8 for m in range(nt) if epi_at[m] == X)

        out = []
        for S in range(nst):
            y = KCPW * min(nst - 1 - S, KCD - 2) + sum(visible_at(X) for X in range(max(0, S - KCD + 2), S + 1))
            if nst1 <= S <= nst1 + KCD - 2:
                y += 2 * nt
            out.append(y)
        return out

    text = open(_asm("wn_col2.hip", tmp_path)).read()
    seen = 0
    for m in re.finditer(r"^(_ZN2wn12hcol2_kernel\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        bf, nt, hasdr = re.search(r"hcol2_kernel<(\w+), (\d+), (\w+)>", name).groups()
        lines = [l.strip() for l in m.group(2).splitlines()]
        bars = [i for i, l in enumerate(lines) if l == "s_barrier"]
        waits = []
        for b in bars[1:]:
            w = [l for l in lines[max(0, b - 3):b] if re.match(r"s_waitcnt vmcnt\(\d+\)$", l)]
            assert w, (name, "no counted wait in front of a K-loop barrier")
            waits.append(int(re.match(r"s_waitcnt vmcnt\((\d+)\)", w[-1]).group(1)))
        assert waits == expected(int(nt), hasdr == "true"), (name, waits, expected(int(nt), hasdr == "true"))
        seen += 1
    assert seen == 16, seen
