#!/usr/bin/env python
"""BASELINE configs[4] on its real input format: `regenie-amd --step 2 --qt --bgen` on a BGEN v1.2 file (layout 2, 8 bits per probability,
zlib) written here at the full sample count -- 500,000 samples x M variants, 10 quantitative phenotypes, 10 covariates -- from process
start to exit, with the shares of the run the driver's log gives (inflate + probabilities -> dosages on the host threads, PCIe, device),
next to regenie itself (oracle/_ref/regenie, when present) on a BOUNDED sample of the same data: a second BGEN file with the first
`ref_variants` variants, on which both programs run and their result lines are compared.

The LOCO predictions both programs read come from this driver's own step 1 on a small .bed of the same samples.
Usage (GPU box):  python tools/bgen_e2e.py [--json record.json] [N=500000] [M=100000] [ref_variants=2000] [P=10] [C=10]
The .bgen is written by a pool of worker processes (one zlib stream per variant; 1.5 MB each at 500,000 samples)."""
import multiprocessing as mp
import os
import struct
import subprocess
import sys
import time
import zlib

import numpy as np

D = "/tmp/bgen_e2e"


def _variant_block(args):
    """One variant's identifying data + compressed genotype block (published BGEN v1.2 layout): hard calls from Binomial(2, maf) with 30 %
    of them smeared into genuine probabilities, 0.2 % missing."""
    j, n, chrom = args
    rng = np.random.default_rng(1000003 + j)
    maf = 0.02 + 0.48 * rng.random()
    g = (rng.random(n) < maf).astype(np.uint8) + (rng.random(n) < maf).astype(np.uint8)       # copies of the FIRST allele
    hom = np.where(g == 2, 255, 0).astype(np.int16)
    het = np.where(g == 1, 255, 0).astype(np.int16)
    sm = rng.random(n) < 0.3
    a = (rng.random(n) * 80).astype(np.int16)
    hom = np.where(sm & (g == 2), hom - a, hom)
    het = np.where(sm & (g == 2), het + a // 2, het)
    het = np.where(sm & (g == 1), het - a, het)
    hom = np.where(sm & (g == 1), hom + a // 2, hom)
    het = np.where(sm & (g == 0), het + a // 2, het)
    miss = rng.random(n) < 0.002
    probs = np.stack([np.where(miss, 0, hom), np.where(miss, 0, het)], axis=1).astype(np.uint8)
    blk = struct.pack("<IHBB", n, 2, 2, 2) + np.where(miss, 0x82, 0x02).astype(np.uint8).tobytes() + bytes([0, 8]) + probs.tobytes()
    z = zlib.compress(blk, 1)
    rec = bytearray()
    for s in ("", "v%d" % j, str(chrom)):
        b = s.encode()
        rec += struct.pack("<H", len(b)) + b
    rec += struct.pack("<IH", j + 1, 2)
    for al in ("A", "G"):
        rec += struct.pack("<I", 1) + al.encode()
    rec += struct.pack("<II", len(z) + 4, len(blk)) + z
    return bytes(rec)


def write_bgen(path, n, m, chroms, nproc):
    flags = 1 | (2 << 2)                                  # zlib, layout 2, no embedded sample identifiers
    header = struct.pack("<III", 20, m, n) + b"bgen" + struct.pack("<I", flags)
    t0 = time.time()
    with open(path, "wb") as fh, mp.Pool(nproc) as pool:
        fh.write(struct.pack("<I", len(header)) + header)
        for rec in pool.imap(_variant_block, ((j, n, chroms[j]) for j in range(m)), chunksize=8):
            fh.write(rec)
    return time.time() - t0


def run(N=500000, M=100000, MREF=2000, P=10, C=10, bsizes=(400, 1000), ref_threads=(16, 64), say=print):
    """Writes the files, runs both programs, returns the record (bench.py's `step2.bgen_from_file`) and prints the lines of the log."""
    import re
    import shutil
    rec = {"samples": N, "variants": M, "phenotypes": P, "covariates": C, "encoding": "BGEN v1.2, layout 2, 8-bit probabilities, zlib", "runs": []}
    shutil.rmtree(D, ignore_errors=True)
    os.makedirs(D, exist_ok=True)
    nproc = max(1, min(224, (os.cpu_count() or 8) - 8))
    rng = np.random.default_rng(5)
    # sample files, phenotypes, covariates
    with open(D + "/x.sample", "w") as fh:
        fh.write("ID_1 ID_2 missing\n0 0 0\n" + "".join("%d %d 0\n" % (i + 1, i + 1) for i in range(N)))
    cov = rng.standard_normal((N, C))
    y = rng.standard_normal((N, P)) + 0.2 * cov[:, :1]
    with open(D + "/x.pheno", "w") as f2, open(D + "/x.covar", "w") as f3:
        f2.write("FID IID " + " ".join("Y%d" % (q + 1) for q in range(P)) + "\n")
        f3.write("FID IID " + " ".join("C%d" % (q + 1) for q in range(C)) + "\n")
        f2.write("".join("%d %d " % (i + 1, i + 1) + " ".join("%.6f" % v for v in y[i]) + "\n" for i in range(N)))
        f3.write("".join("%d %d " % (i + 1, i + 1) + " ".join("%.6f" % v for v in cov[i]) + "\n" for i in range(N)))
    # a small .bed of the same samples for step 1 (LOCO predictions for both programs)
    ms1 = 2200
    with open(D + "/s.bed", "wb") as fh:
        fh.write(b"\x6c\x1b\x01")
        fh.write(rng.integers(0, 256, size=(ms1, N // 4), dtype=np.uint8).tobytes())
    with open(D + "/s.bim", "w") as fh:
        fh.write("".join("%d\ts%d\t0\t%d\tA\tG\n" % (j // 100 + 1, j, j + 1) for j in range(ms1)))
    with open(D + "/s.fam", "w") as fh:
        fh.write("".join("%d %d 0 0 0 -9\n" % (i + 1, i + 1) for i in range(N)))
    nchr = int(os.environ.get("BGEN_E2E_NCHR", "22"))       # chromosomes the variants are spread over (10 M variants over 22: 450,000 each; a bounded
    chroms = [j * nchr // M + 1 for j in range(M)]           # sample keeps whole blocks per chromosome with fewer of them)
    rec["chromosomes"] = nchr
    tw = write_bgen(D + "/x.bgen", N, M, chroms, nproc)
    rec["file_gb"] = round(os.path.getsize(D + "/x.bgen") / 1e9, 2)
    say("x.bgen: %d samples x %d variants, %.1f GB, written in %.0f s by %d processes" % (N, M, rec["file_gb"], tw, nproc))
    tw = write_bgen(D + "/r.bgen", N, MREF, [j * nchr // MREF + 1 for j in range(MREF)], nproc)
    say("r.bgen: the reference's bounded sample, %d variants, %.2f GB" % (MREF, os.path.getsize(D + "/r.bgen") / 1e9))
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "..", "regenie_amd", "bin", "regenie-amd")
    ref = os.path.join(here, "..", "oracle", "_ref", "regenie")
    t0 = time.time()
    r = subprocess.run([exe, "--step", "1", "--qt", "--bed", D + "/s", "--phenoFile", D + "/x.pheno", "--covarFile", D + "/x.covar", "--bsize", "100",
                        "--out", D + "/s1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    say("step 1 (LOCO predictions, %d SNPs): %.1f s" % (ms1, time.time() - t0))
    common = ["--step", "2", "--qt", "--sample", D + "/x.sample", "--phenoFile", D + "/x.pheno", "--covarFile", D + "/x.covar", "--pred", D + "/s1_pred.list"]
    cpu_max = open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a"
    rec["host"] = {"hardware_threads": os.cpu_count(), "cgroup_cpu_max": cpu_max}
    say("host: %s hardware threads, cpu.max %s" % (os.cpu_count(), cpu_max))
    if os.environ.get("BGEN_E2E_BSIZES"):
        bsizes = tuple(int(b) for b in os.environ["BGEN_E2E_BSIZES"].split(","))
    variants = [("bsize %d" % b, b, {}) for b in bsizes]
    for extra in os.environ.get("BGEN_E2E_VARIANTS", "").split(";"):          # e.g. "RG_S2_BGEN_HOST=1;RG_S2_PREP_THREADS=64,RG_BGEN_ZLIB=1"
        if extra:
            variants.append(("bsize %d %s" % (bsizes[0], extra), bsizes[0], dict(kv.split("=") for kv in extra.split(","))))
    for name, bsz, env in variants:
        t0 = time.time()
        r = subprocess.run([exe] + common + ["--bgen", D + "/x.bgen", "--bsize", str(bsz), "--out", D + "/s2"] + os.environ.get("BGEN_E2E_ARGS", "").split(), capture_output=True, text=True,
                           env=dict(os.environ, RG_TIMING="1", **env))
        dt = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        marks = [ln.strip() for ln in (r.stdout + r.stderr).split("\n") if "Elapsed" in ln or "since start" in ln or "[timing] step 2" in ln or "[timing] chromosome" in ln]
        chr_ms = [int(x) for x in re.findall(r"reading loco predictions for the chromosome\.\.\.done \((\d+)ms\)", r.stdout)]
        blk_ms = [int(x) for x in re.findall(r"block \[\d+/\d+\] : done \((\d+)ms\)", r.stdout)]
        run = {"name": name, "bsize": bsz, "wall_s": round(dt, 2), "variants_per_s": round(M / dt, 1), "variant_sample_pheno_per_s": M * N * P / dt}
        tm = re.search(r"read-ahead (\d+)\) \| chromosome set-up (\d+) ms \| waiting for the prepared block (\d+) ms \(preparing: (\d+) ms wall, overlapped; (\d+) thread-ms inflate "
                       r"\+ (\d+) thread-ms byte walk\) \| upload \+ device \+ results (\d+) ms \| formatting \+ writing (\d+) ms", r.stderr)
        td = re.search(r"BGEN on the device: (\d+) blocks \((\d+) on the host route\) \| reading the stored streams (\d+) ms \| copy \+ inflate \+ walk on the GPU (\d+) ms", r.stderr)
        if td:
            run["bgen_on_device"] = {"blocks": int(td.group(1)), "blocks_on_host_route": int(td.group(2)), "read_streams_ms": int(td.group(3)),
                                     "copy_inflate_walk_ms": int(td.group(4))}
        if tm:
            v = [int(x) for x in tm.groups()]
            run["shares_ms"] = {"host_threads_read_ahead": v[0], "chromosome_setup": v[1], "waiting_for_prepared_block": v[2], "prepare_wall_overlapped": v[3],
                                "inflate_thread_ms": v[4], "byte_walk_thread_ms": v[5], "upload_device_results": v[6], "format_write": v[7]}
            # what 10 M variants over 22 chromosomes approach: the slower of the read-ahead (inflate + byte walk) and the main thread's share of a block
            run["variants_per_s_block_loop"] = round(M / max(1e-9, max(v[3], v[2] + v[6] + v[7]) / 1e3), 1)
        if chr_ms and blk_ms:
            marks.append("chromosome set-ups %d x median %d ms (sum %d); blocks %d x median %d ms (sum %d)"
                         % (len(chr_ms), sorted(chr_ms)[len(chr_ms) // 2], sum(chr_ms), len(blk_ms), sorted(blk_ms)[len(blk_ms) // 2], sum(blk_ms)))
        rec["runs"].append(run)
        say("regenie-amd --step 2 --bgen, %-28s: wall %.1f s = %.0f variants/s = %.2e variant*sample*pheno/s from the file | %s"
            % (name, dt, M / dt, M * N * P / dt, " | ".join(marks)))
    # hard calls: both programs on the small .bed itself (2,200 variants x N samples, the LOCO predictions above) -- regenie's rate there is the
    # `cpu_baseline` of bench.py's Step-2 record (kind "reference")
    if os.path.exists(ref):
        bed_args = ["--step", "2", "--qt", "--bed", D + "/s", "--phenoFile", D + "/x.pheno", "--covarFile", D + "/x.covar", "--pred", D + "/s1_pred.list", "--bsize", "400"]
        t0 = time.time()
        r = subprocess.run([exe] + bed_args + ["--out", D + "/b_amd"], capture_output=True, text=True)
        t_amd_bed = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        thr = ref_threads[0]
        t0 = time.time()
        r = subprocess.run([ref] + bed_args + ["--threads", str(thr), "--out", D + "/b_ref"], capture_output=True, text=True)
        dt = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        same = tot = 0
        for q in range(P):
            a = open(D + "/b_amd_Y%d.regenie" % (q + 1)).read().splitlines()
            b = open(D + "/b_ref_Y%d.regenie" % (q + 1)).read().splitlines()
            assert len(a) == len(b)
            tot += len(b) - 1
            same += sum(x == z for x, z in zip(a[1:], b[1:]))
        rec["bed_reference"] = {"kind": "reference", "program": "regenie v4.1.2 (oracle/_ref)", "threads": thr, "variants": ms1, "wall_s": round(dt, 2),
                                "variants_per_s": round(ms1 / dt, 1), "value": ms1 * N * P / dt, "unit": "variant*sample*pheno/s", "regenie_amd_wall_s": round(t_amd_bed, 2),
                                "result_lines": tot, "byte_identical": same,
                                "sample": "regenie --step 2 --qt --bed on %d variants x %d samples x %d phenotypes (hard calls, 25 %% of them missing), process start to exit" % (ms1, N, P)}
        say("hard calls, %d variants x %d phenotypes from a .bed: regenie v4.1.2 (oracle/_ref, --threads %d) %.1f s = %.0f variants/s; regenie-amd %.1f s; %d of %d result lines byte-identical"
            % (ms1, P, thr, dt, ms1 / dt, t_amd_bed, same, tot))
    # the bounded sample: both programs, line by line
    t0 = time.time()
    r = subprocess.run([exe] + common + ["--bgen", D + "/r.bgen", "--bsize", str(bsizes[0]), "--out", D + "/r_amd"] + os.environ.get("BGEN_E2E_ARGS", "").split(),      # e.g. "--gpus 8"
                       capture_output=True, text=True)
    t_amd = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if os.path.exists(ref):
        for thr in ref_threads:
            t0 = time.time()
            r = subprocess.run([ref] + common + ["--bgen", D + "/r.bgen", "--bsize", "400", "--threads", str(thr), "--out", D + "/r_ref"], capture_output=True, text=True)
            dt = time.time() - t0
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            same = close = tot = 0
            for q in range(P):
                a = open(D + "/r_amd_Y%d.regenie" % (q + 1)).read().splitlines()
                b = open(D + "/r_ref_Y%d.regenie" % (q + 1)).read().splitlines()
                assert len(a) == len(b)
                tot += len(b) - 1
                same += sum(x == z for x, z in zip(a[1:], b[1:]))
                num = lambda ln: [t for k, t in enumerate(ln.split()) if k in (5, 6, 7, 9, 10, 11, 12)]     # A1FREQ INFO N | BETA SE CHISQ LOG10P
                close += sum(all(abs(float(u) - float(v)) <= 2e-5 * max(abs(float(v)), 1e-3) for u, v in zip(num(x), num(z)) if u != "NA" and v != "NA")
                             for x, z in zip(a[1:], b[1:]))
            rec.setdefault("cpu_baseline", []).append({"kind": "reference", "program": "regenie v4.1.2 (oracle/_ref)", "threads": thr, "sample": "%d variants of the same encoding" % MREF,
                                                       "wall_s": round(dt, 1), "variants_per_s": round(MREF / dt, 1), "regenie_amd_wall_s": round(t_amd, 1),
                                                       "result_lines": tot, "byte_identical": same, "within_2e-5": close})
            say("bounded sample (%d variants x %d phenotypes): regenie v4.1.2 (oracle/_ref, --threads %d) %.1f s = %.0f variants/s; regenie-amd %.1f s; "
                "%d of %d result lines byte-identical, %d within 2e-5" % (MREF, P, thr, dt, MREF / dt, t_amd, same, tot, close))
    shutil.rmtree(D, ignore_errors=True)
    return rec


def main(argv):
    """[--json PATH] [N M MREF P C]; with --json the record goes to PATH (bench.py's sub-run: one bsize, one reference thread count)."""
    js = None
    if argv and argv[0] == "--json":
        js, argv = argv[1], argv[2:]
    kw = dict(bsizes=(400,), ref_threads=(16,)) if js else {}
    rec = run(*[int(a) for a in argv], say=lambda s: print(s, flush=True), **kw)
    if js:
        import json
        with open(js, "w") as fh:
            json.dump(rec, fh)


if __name__ == "__main__":
    main(sys.argv[1:])
