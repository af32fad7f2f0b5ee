"""One training step of bench.py as a kernel timeline (rocprofv3 --kernel-trace): every launch between two optimizer steps with its
start offset, duration and the idle gap before it.  Usage (GPU box): bash tools/run_step_timeline.sh -> gpurun_out/step_timeline.txt"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"at::native::", "", name)
    return name[:100]


def main(d, out):
    rows = []
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # one step = from the end of the last optimizer launch of a step to the end of the last one of the next (a step with several
    # learning-rate segments launches adamw_kernel once per segment, back to back behind ONE sum-of-squares launch pair: sumsq_partial_kernel + sumsq_fold_kernel)
    opt = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    opt = [i for k, i in enumerate(opt) if k + 1 == len(opt) or any("sumsq_" in rows[j][2] for j in range(i, opt[k + 1]))]
    a, b = opt[-2], opt[-1]
    t0 = rows[a][1]
    with open(out, "w") as fh:
        fh.write("step: %d launches, %.3f ms\n" % (b - a, (rows[b][1] - t0) / 1e6))
        prev = t0
        busy = 0
        for s, e, n in rows[a + 1:b + 1]:
            busy += e - s
            fh.write("%9.1f %8.1f %6.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short(n)))
            prev = e
        fh.write("busy %.3f ms\n" % (busy / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
