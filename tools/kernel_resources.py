#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(compile-time facts for gfx950; no GPU needed).  Usage: python tools/kernel_resources.py > profiles/r1_kernel_resources.md"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from regenie_amd import build as b  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o.replace("(anonymous namespace)::", "")).replace("void ", "") for o in out]


def main():
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for src in b.SOURCES:
            if not src.endswith(".hip"):
                continue
            flags = [f for f in b.FLAGS] + ["-w", "-Rpass-analysis=kernel-resource-usage"]
            r = subprocess.run([b._hipcc()] + flags + ["-c", os.path.join(b.CSRC, src), "-o", os.path.join(d, "x.o")], capture_output=True, text=True)
            cur = None
            for ln in r.stderr.split("\n"):
                m = re.search(r"remark: Function Name: (\S+)", ln)
                if m:
                    cur = {"file": src, "name": m.group(1)}
                    rows.append(cur)
                    continue
                m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", ln)
                if m and cur is not None:
                    cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    print("# Kernel resource usage (gfx950, hipcc -O3; `python tools/kernel_resources.py`)\n")
    print("| file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B/block | waves/SIMD |")
    print("|---|---|---:|---:|---:|---:|---:|---:|")
    for r, n in zip(rows, names):
        print("| %s | `%s` | %s | %s | %s | %s | %s | %s |" % (r["file"], n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                                               r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"),
                                                               r.get("Occupancy [waves/SIMD]")))
    spills = [n for r, n in zip(rows, names) if r.get("ScratchSize [bytes/lane]") not in ("0", None)]
    print("\nKernels with scratch: %s" % (", ".join(spills) if spills else "none"))


if __name__ == "__main__":
    main()
