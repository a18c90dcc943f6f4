#!/usr/bin/env python
"""Per-dispatch view of a rocprofv3 --kernel-trace CSV: for the LAST level-0 batch sequence, prints each
Cholesky launch (name, grid, duration) and per-kernel / per-phase totals.  Usage: trace_seq.py <dir> [out.md]"""
import csv
import glob
import sys


def main(d, out=None):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # find the index of the last k_gram_fp4_blocks / k_gram_blocks with a full-size grid, take launches up to k_l0_scale
    def nm(r):
        return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].replace("void ", "").strip()
    starts = [i for i, r in enumerate(rows) if nm(r) == "k_bed_prep_rows"]
    # choose the first batch of the last step: batches of a step are consecutive k_bed_prep_rows; pick 4th from the end
    i0 = starts[-4] if len(starts) >= 4 else starts[0]
    i1 = starts[-3] if len(starts) >= 4 else len(rows)
    lines = ["| # | kernel | grid | wg | us |", "|---:|---|---:|---:|---:|"]
    agg = {}
    t_first, t_last = int(rows[i0]["Start_Timestamp"]), int(rows[i1 - 1]["End_Timestamp"])
    for k, r in enumerate(rows[i0:i1]):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"])
        lines.append("| %d | %s | %d | %d | %.1f |" % (k, nm(r), g // wg, wg, us))
        a = agg.setdefault(nm(r), [0, 0.0])
        a[0] += 1
        a[1] += us
    lines.append("")
    lines.append("batch wall (first start to last end): %.1f us; sum of kernel durations: %.1f us" %
                 ((t_last - t_first) / 1e3, sum(a[1] for a in agg.values())))
    lines.append("")
    lines.append("| kernel | launches | total us |")
    lines.append("|---|---:|---:|")
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.1f |" % (n, c, us))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
