#!/usr/bin/env python3
"""Scan the gfx950 assembly of the LDS-DMA ring kernels for a compiler-inserted `s_waitcnt vmcnt(0)` inside the K loop.

hipcc may decide that an LDS read could alias a pending LDS-DMA and drain the whole ring in front of it (it did in
hwgrad_kernel).  The K loop of every hgemm / hwgrad instantiation must have its counted
`s_waitcnt vmcnt(N)`, N > 0, right before its s_barrier (N = 0 only in the two-stage f16x3 wgrad ring) and no vmcnt(0)
between the barrier and the last MFMA.
usage: hipcc ... -save-temps=obj ; check_rings.py <file.s> [...]"""
import re, subprocess, sys

bad = 0
for path in sys.argv[1:]:
    s = open(path).read()
    for m in re.finditer(r"^(_ZN2wn\w*(?:hgemm_kernel|hgemm8_kernel|hwgrad_kernel|hfused_fwd_kernel)\w*):[^\n]*\n(.*?)\n\.Lfunc_end", s, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        lines = [l.strip() for l in m.group(2).splitlines()]
        bars = [i for i, l in enumerate(lines) if l == "s_barrier"]
        if not bars:
            print("BAD  no barrier in", name); bad += 1; continue
        # K-loop segments = the code between a barrier and the next one (or the end) that contains MFMAs
        segs = []
        for bi, b in enumerate(bars):
            e = bars[bi + 1] if bi + 1 < len(bars) else len(lines)
            body = lines[b + 1:e]
            mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
            if not mf:
                continue
            counted = [l for l in lines[max(0, b - 4):b] if re.match(r"s_waitcnt vmcnt\(\d+\)", l)]
            n = int(re.match(r"s_waitcnt vmcnt\((\d+)\)", counted[-1]).group(1)) if counted else -1
            drains = [l for l in body[:mf[-1] + 1] if re.match(r"s_waitcnt.*vmcnt\(0\)", l)]
            segs.append((n, len(mf), len(drains)))
        # the f16x3 wgrad ring and the 256 x 256 f16x3 GEMM have two 64 KiB stages: their wait at the barrier is vmcnt(0) by design (the next stage is issued
        # AFTER the barrier); everywhere else a prefetch must stay in flight across the barrier
        two_stage = "hwgrad_kernel<2," in name or re.search(r"hgemm8_kernel<2, \w+, \d, 8, 4>", name) is not None
        ok = bool(segs) and all((n > 0 or (two_stage and n == 0)) and d == 0 for n, _, d in segs)
        bad += 0 if ok else 1
        n, mf, drains = (segs[0][0], [0] * sum(x[1] for x in segs), [0] * sum(x[2] for x in segs)) if segs else (-1, [], [])
        print("%-4s %-72s vmcnt(%d) before the barrier, %d MFMAs, %d vmcnt(0) inside the K loop" % ("ok" if ok else "BAD", name[:72], n, len(mf), len(drains)))
sys.exit(1 if bad else 0)
