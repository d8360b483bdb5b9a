#!/usr/bin/env python3
"""Scan the gfx950 assembly hipcc left under jittor-mlp_amd/build/ for vector-memory loads that are issued right behind an
`s_waitcnt vmcnt(0)` INSIDE a loop: the signature of loads behind a branch (`if (ok) r = *p;`), which hipcc serialises -- one
memory latency per load -- as found in the depthwise kernel (profiles/r04_dwconv_variants.txt).
usage: python tools/scan_serial_loads.py [min_count]"""
import glob, os, re, sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jittor-mlp_amd", "build")
mn = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows = []
for path in sorted(glob.glob(os.path.join(root, "*gfx950*.s"))):
    name, body = None, []
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+|mlpk\w+):\s", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            inloop, serial, loads, lastwait = False, 0, 0, -10
            for k, l in enumerate(body):
                if "Loop Header" in l or "in Loop:" in l:
                    inloop = True
                t = l.strip()
                if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                    lastwait = k
                if re.match(r"(global_load|buffer_load|flat_load)", t):
                    if inloop:
                        loads += 1
                        # nothing but address arithmetic between the wait and the load
                        if k - lastwait <= 4:
                            serial += 1
            if serial >= mn:
                rows.append((serial, loads, os.path.basename(path).split("-hip")[0], name[:110]))
            name = None
            continue
        body.append(line)
for r in sorted(rows, reverse=True):
    print("%4d of %4d loads in loops behind vmcnt(0)   %-18s %s" % r)
