#!/usr/bin/env python
"""Round 6: BASELINE configs[2] from files with the whole driver log kept (tools/cli_e2e.py prints marks only).  Writes the 62.5 GB .bed once
(or reuses /tmp/e2e/x.bed of the right size), then runs `regenie-amd --step 1` under the environments given on the command line:
    python tools/r6_ingest.py "NAME=VALUE,NAME=VALUE" "..." ...      ("" = defaults); logs -> gpurun_out/r6_ingest/run<k>.log"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    N, M, P = 500000, 500000, 10
    d = "/tmp/e2e"
    want = 3 + M * (N // 4)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "r6_ingest")
    os.makedirs(out, exist_ok=True)
    if not (os.path.exists(d + "/x.bed") and os.path.getsize(d + "/x.bed") == want):
        import tools.cli_e2e as ce
        # generate through the tool's own writer, without its runs: it writes the data set first, then runs variants -- stop it after one
        os.environ["RG_E2E_WRITE_ONLY"] = "1"
        ce.main(N, M, P)
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "regenie_amd", "bin", "regenie-amd")
    for k, spec in enumerate(sys.argv[1:] or [""]):
        env = dict(os.environ)
        for kv in [s for s in spec.split(",") if s]:
            a, b = kv.split("=", 1)
            if a == "SLEEP":            # let the runtime finish scrubbing what the previous process released
                time.sleep(float(b))
                continue
            env[a] = b
        t0 = time.time()
        r = subprocess.run([exe, "--step", "1", "--bed", d + "/x", "--phenoFile", d + "/x.pheno", "--covarFile", d + "/x.covar", "--bsize", "1000",
                            "--out", d + "/out"], capture_output=True, text=True, env=env)
        dt = time.time() - t0
        open(os.path.join(out, "run%d.log" % k), "w").write("# %s\n# wall %.2f s rc %d\n" % (spec, dt, r.returncode) + r.stdout + r.stderr)
        marks = [ln.strip() for ln in (r.stdout + r.stderr).split("\n") if "since start" in ln]
        print("[%s] wall %.2f s rc %d | %s" % (spec, dt, r.returncode, " | ".join(marks)), flush=True)


if __name__ == "__main__":
    main()
