#!/usr/bin/env python3
"""Condense hipcc -Rpass-analysis=kernel-resource-usage remarks into one line per kernel."""
import re
import subprocess
import sys

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = re.sub(r"pcg::|\(pcg::StepArgs\)|\(pcg::DevConst const\*.*\)|void ", "", r["name"])
    print(f"{n:70s} vgpr={r.get('VGPRs','?'):>4s} agpr={r.get('AGPRs','?'):>3s} sgpr={r.get('SGPRs','?'):>4s} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>5s} occ={r.get('Occupancy [waves/SIMD]','?')} lds={r.get('LDS Size [bytes/block]','?')}")
