#!/usr/bin/env python
"""List scratch spills / reloads of a kernel and which of them sit inside a loop that issues MFMAs (the K loop).
usage: spillcheck.py file.s <substring of the kernel symbol>"""
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# basic blocks: label line -> (depth, header) from the compiler's loop comments
blocks, cur = [], None
for i, l in enumerate(body):
    if l.startswith(".LBB") or l.startswith("; %bb."):
        m = re.search(r"Depth=(\d+)", l)
        cur = {"start": i, "depth": int(m.group(1)) if m else 0, "lines": []}
        blocks.append(cur)
    elif cur is not None:
        if "Depth=" in l and l.strip().startswith(";"):
            m = re.search(r"Depth=(\d+)", l)
            cur["depth"] = max(cur["depth"], int(m.group(1)))
        cur["lines"].append((i, l))
sp = sum("Folded Spill" in l for l in body)
rl = sum("Folded Reload" in l for l in body)
hot = 0
for b in blocks:
    if b["depth"] >= 2 and any("v_mfma" in l for _, l in b["lines"]):
        hot += sum("Folded Reload" in l or "Folded Spill" in l for _, l in b["lines"])
print("%s: %d lines, spills %d, reloads %d, spill/reload instructions inside MFMA loops (depth>=2): %d, mfma %d" %
      (needle, len(body), sp, rl, hot, sum("v_mfma" in l for l in body)))
