#!/usr/bin/env python3
"""Print the order of DMA pieces (D), loads (L), stores (S), barriers (|), MFMAs (m), LDS reads (r) and every s_waitcnt of
the hcol_kernel instantiations named on the command line (substrings of the demangled name), from an assembly listing."""
import re, subprocess, sys
t = open(sys.argv[1]).read()
for n in re.findall(r"^(_ZN2wn11hcol_kernel\w+):", t, re.M):
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    if not any(w in d for w in sys.argv[2:]):
        continue
    i = t.index(n + ":"); b = t[i:t.index("s_endpgm", i)]
    seq = []
    for line in b.splitlines():
        l = line.strip()
        if l.startswith("s_waitcnt"): seq.append(l.replace("s_waitcnt ", ""))
        elif l.startswith("global_load_lds"): seq.append("D")
        elif l.startswith("global_load"): seq.append("L")
        elif l.startswith("global_store"): seq.append("S")
        elif l.startswith("s_barrier"): seq.append("|")
        elif l.startswith("v_mfma"): seq.append("m")
        elif l.startswith("ds_read"): seq.append("r")
        elif l.startswith("s_cbranch") or l.startswith("s_branch"): seq.append("BR")
    out, prev, cnt = [], None, 0
    for x in seq + [None]:
        if x == prev: cnt += 1
        else:
            if prev is not None: out.append(prev + (str(cnt) if cnt > 1 else ""))
            prev, cnt = x, 1
    print(d); print(" ".join(out)); print()
