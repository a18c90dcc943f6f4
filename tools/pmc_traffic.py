#!/usr/bin/env python
"""HBM traffic of the Cholesky kernel group from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected separately: they do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots').
rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads
(HBM section of the same guide), so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is taken as reported.
Usage: pmc_traffic.py <fetch_dir> <write_dir> <n_level0_batches_in_the_run> <out.json> [blocks phenos]
The file records the digests of the sources it was measured on (regenie_amd/lib/build.stamp: everything; lib/library.stamp: the kernel
library without the host driver); bench.py refuses a file whose library digest is not the current build's."""
import csv
import glob
import json
import os
import sys

GROUP = ("k_c128_panel", "k_chol_update", "k_chol_gfact", "k_chol_gstrip", "k_chol_backsolve", "k_chol_backsolve_mfma")      # (k_c128_panel: chol_p128.h, round 6)


def total(d, counter, level0_only=True):
    tot = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].replace("void ", "").strip()
            tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"])
            COUNT[(counter, name)] = COUNT.get((counter, name), 0) + 1
    return tot


COUNT = {}


GROUPS = {"chol": GROUP, "l1_gram": ("k_l1_gram128", "k_l1_gram64", "k_l1_wty", "k_reduce_slices", "k_sum_folds"),
          "gram_fp4": ("k_gram_fp4_blocks",), "pred": ("k_pk_transpose", "k_beta_split", "k_l0_pred_i8", "k_l0_pred", "k_beta_post",
                                                       "k_l0_stats", "k_l0_scale"),
          "irls_stream": ("k_bt_eval", "k_bt_score"),
          "wgram": ("k_wgram_mx", "k_wgram_bf16", "k_wsplit", "k_sqrtw", "k_wgram128", "k_wg_reduce", "k_wg_wz", "k_wgram")}


def main(fd, wd, nbatch, out, blocks=None, phenos=None):
    nbatch = int(nbatch)
    f, w = total(fd, "FETCH_SIZE"), total(wd, "WRITE_SIZE")
    rd = sum(f.get(k, 0.0) for k in GROUP) * 1024 * 2
    wr = sum(w.get(k, 0.0) for k in GROUP) * 1024
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        stamp = open(os.path.join(here, "..", "regenie_amd", "lib", "build.stamp")).read().strip()
    except OSError:
        stamp = None
    try:
        lib_stamp = open(os.path.join(here, "..", "regenie_amd", "lib", "library.stamp")).read().strip()
    except OSError:
        lib_stamp = None
    try:
        file_digests = json.load(open(os.path.join(here, "..", "regenie_amd", "lib", "kernel_files.json")))
    except (OSError, ValueError):
        file_digests = None
    groups = {}
    for gname, ks in GROUPS.items():
        grd = sum(v for k, v in f.items() if k.split("<")[0] in ks) * 1024 * 2
        gwr = sum(v for k, v in w.items() if k.split("<")[0] in ks) * 1024
        lead = next((k for k in ks if ("FETCH_SIZE", k) in COUNT), ks[0])      # dispatches of the group's leading kernel in the read pass
        groups[gname] = {"read_bytes": grd, "write_bytes": gwr, "hbm_bytes": grd + gwr, "lead_kernel": lead, "lead_launches": COUNT.get(("FETCH_SIZE", lead), 0),
                         "group_launches": sum(n for (c, k), n in COUNT.items() if c == "FETCH_SIZE" and k.split("<")[0] in ks)}
    res = {"kernel_group": list(GROUP), "level0_batches": nbatch, "build_stamp": stamp, "library_stamp": lib_stamp, "source_digests": file_digests, "groups": groups,
           "blocks": int(blocks) if blocks else None, "phenos": int(phenos) if phenos else None,
           "read_bytes_per_batch": rd / nbatch, "write_bytes_per_batch": wr / nbatch,
           "hbm_bytes_per_batch": (rd + wr) / nbatch,
           "note": "FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, summed over the Cholesky kernels of one run "
                   "(level-0 and level-1 launches) and divided by the number of level-0 batches",
           "per_kernel_read_GB": {k: f.get(k, 0.0) * 2048 / 1e9 for k in f if k.startswith("k_")},
           "per_kernel_write_GB": {k: w.get(k, 0.0) * 1024 / 1e9 for k in w if k.startswith("k_")}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:600])


if __name__ == "__main__":
    main(*sys.argv[1:7])
