#!/usr/bin/env python
"""Condenses a rocprofv3 `--kernel-trace --stats --output-format csv` run into a short per-kernel table
(kernel name cut at the first '(' / '<'), written as markdown for profiles/."""
import csv
import glob
import sys


def main(d, out=None, steps=1):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f:
        print("no kernel_stats.csv under", d)
        return
    rows = list(csv.DictReader(open(f[0])))
    agg = {}
    for r in rows:
        name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].replace("void ", "").strip()
        if not name.startswith(("k_", "__amd", "Cijk")):
            name = "torch/other"
        a = agg.setdefault(name, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    tot = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for name, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.3f | %.1f | %.1f |" % (name, calls, ns / 1e6, ns / calls / 1e3, 100 * ns / tot))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
