#!/usr/bin/env python3
"""ISA mnemonic histogram of one kernel in a hipcc -S output: isa_hist.py file.s <mangled-name-substring>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [l.split(":")[0] for l in s.splitlines() if re.match(r"^[A-Za-z_][\w$.]*:", l) and pat in l.split(":")[0]]
for name in names[: int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    i = s.index("\n" + name + ":")
    j = s.index(".Lfunc_end", i)
    ins = [l.strip().split()[0] for l in s[i:j].splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    print(name, "total", len(ins))
    print("  " + "  ".join(f"{k}:{v}" for k, v in c.most_common(40)))
