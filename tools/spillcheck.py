#!/usr/bin/env python
"""List scratch spills / reloads of a kernel and whether they sit inside its innermost (Depth=2) loop.
usage: spillcheck.py file.s <substring of the kernel symbol>"""
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and l.rstrip().endswith(":") is False and ":" in l and l.split(":")[0].find(needle) >= 0)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
depth2 = [i for i, l in enumerate(body) if "Depth=2" in l]
lo, hi = (min(depth2), max(depth2)) if depth2 else (-1, -1)
# extend hi to the end of the last depth-2 block (next label)
for i in range(hi + 1, len(body)):
    if body[i].startswith(".LBB") and "Depth=2" not in body[i]:
        hi = i
        break
sp = [i for i, l in enumerate(body) if "Folded Spill" in l]
rl = [i for i, l in enumerate(body) if "Folded Reload" in l]
inl = [i for i in rl if lo <= i <= hi]
print("%s: %d lines, inner loop %d..%d, spills %d, reloads %d, reloads inside inner loop %d, mfma %d" %
      (needle, len(body), lo, hi, len(sp), len(rl), len(inl), sum("v_mfma" in l for l in body)))
