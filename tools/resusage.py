#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stderr log) per kernel."""
import re, sys
cur = None; rows = []
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"remark: [^:]+:\d+:\d+: (.*) \[-Rpass", line) or re.search(r":\d+:\d+: remark: (.*) \[-Rpass", line) or re.search(r":\d+:\d+:\s+(.*) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":",1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":",1); cur[k.strip()] = v.strip()
for r in rows:
    print(f"{r['name'][:70]:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')} SGPR {r.get('SGPRs','?')}")
