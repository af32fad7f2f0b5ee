#!/usr/bin/env python
"""Condense `MODCR_TEST_REPORT=1 python -m pytest tests -q -m gpu -s` (one `[tol]` line per tolerance check, tests/helpers.py) into
profiles/rNN_tolerance_report*.txt: per test function and bound the worst observed use, then every check.
usage: tol_summary.py <pytest output> <report file> [title]"""
import re
import sys
from collections import OrderedDict


def main(src, dst, title):
    lines = [l.rstrip("\n") for l in open(src, errors="replace")]
    tol = [l for l in lines if "[tol]" in l]
    tail = [l for l in lines if re.search(r"\d+ passed", l)]
    pat = re.compile(r"\[tol\]\s+(\S+)\s+(.*?)\s+(max\|err\|/scale|relative L2)\s+([0-9.e+-]+)\s+bound\s+([0-9.e+-]+)\s+used\s+(\d+)\s*%")
    groups = OrderedDict()
    for l in tol:
        m = pat.search(l)
        if not m:
            continue
        test, what, kind, err, bound, _ = m.groups()
        fn = test.split("[")[0]
        key = (fn, float(bound), kind)
        g = groups.setdefault(key, [0.0, "", 0])
        g[2] += 1
        if float(err) >= g[0]:
            g[0], g[1] = float(err), what.strip()
    with open(dst, "w") as fh:
        fh.write(title + "\n")
        fh.write("command: MODCR_TEST_REPORT=1 python -m pytest tests -q -m gpu -s   (%s; one [tol] line per check, %d checks)\n"
                 % (tail[-1].strip(" =") if tail else "?", len(tol)))
        fh.write("convention: every max|err| check is  max|got - ref| <= bound * max(1, max|ref|);  gradient checks marked 'relative L2' are "
                 "|got - ref|_2 / |ref|_2 <= bound.\nContract bounds (north_star): 1e-3 fp32, 2e-2 bf16.  Bounds above 2e-2 are deep-stack / "
                 "gradient bounds, each <= ~2x its worst observed use.\n\n== per test function and bound: worst use ==\n")
        for (fn, bound, kind), (err, what, n) in sorted(groups.items(), key=lambda kv: -kv[1][0] / kv[0][1]):
            fh.write("  %3.0f%%  %-62s bound %.1e %-14s worst %.2e (%s) checks=%d\n" % (100.0 * err / bound, fn, bound, kind, err, what, n))
        fh.write("\n== every check ==\n")
        for l in tol:
            fh.write(l[l.index("[tol]") - 2:] + "\n" if l.index("[tol]") >= 2 else l + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "Tolerance use of the GPU parity suite")
