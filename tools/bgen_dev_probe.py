#!/usr/bin/env python
"""Rate of the device BGEN decoder (csrc/bgen_inflate.hip) at full sample count: writes M variants of 500,000 samples the way tools/bgen_e2e.py does
(zlib level 1), reads the stored streams, decodes them in batches on the GPU, checks a sample of the inflated blocks against zlib.
Usage (GPU box): python tools/bgen_dev_probe.py [N=500000] [M=2048] [batch=1024]"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bgen_e2e import write_bgen  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    from regenie_amd.bgen import BgenDevice, BgenFile
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    path = "/tmp/bgen_dev_probe.bgen"
    tw = write_bgen(path, N, M, [1] * M, max(1, min(224, (os.cpu_count() or 8) - 8)))
    print("wrote %d variants x %d samples, %.2f GB in %.0f s" % (M, N, os.path.getsize(path) / 1e9, tw), flush=True)
    with BgenFile(path, threads=32) as f, BgenDevice(0) as d:
        if os.environ.get("BGEN_PROBE_MASK"):       # "P,rate": P traits, each missing for `rate` of the samples (the walk's per-trait corrections)
            Pm, rate = os.environ["BGEN_PROBE_MASK"].split(",")
            mask = (np.random.default_rng(3).random((int(Pm), N)) >= float(rate)).astype(np.uint8)
            d.set_samples(N, mask=mask)
            print("per-trait masks: %s traits, %.1f %% of the samples missing for each" % (Pm, 100 * float(rate)), flush=True)
        else:
            d.set_samples(N)
        pin = {}

        def pinned(n):          # page-locked, as the driver's buffers are (rg_host_alloc)
            if pin.get("n", 0) < n:
                pin["t"] = torch.empty(n + n // 8, dtype=torch.uint8).pin_memory()
                pin["n"] = n + n // 8
            return pin["t"].numpy()[:n]
        for rep in range(3):
            t_read = t_dec = 0.0
            for b0 in range(0, M, batch):
                idx = np.arange(b0, min(M, b0 + batch))
                t0 = time.perf_counter()
                comp, off, clen, ulen = f.read_compressed(idx, threads=32, alloc=pinned)
                t1 = time.perf_counter()
                o = __import__("regenie_amd.bgen", fromlist=["RgBgenDevOut"]).RgBgenDevOut()
                st = np.zeros(idx.size, dtype=np.int32)
                o.status = st.ctypes.data
                import ctypes as C
                rc = d.lib.rg_bgen_dev_decode(d.h, 0, idx.size, comp.ctypes.data, comp.size, off.ctypes.data, clen.ctypes.data, ulen.ctypes.data, 0, C.byref(o))
                t2 = time.perf_counter()
                assert rc == 0 and (st == 0).all(), (rc, st[st != 0][:8])
                t_read += t1 - t0
                t_dec += t2 - t1
            print("rep %d: batch %d: read %.3f s (%.1f GB/s), copy + decode + walk %.3f s = %.0f variants/s (%.1f GB/s inflated)"
                  % (rep, batch, t_read, float(clen.sum(dtype=np.float64)) / 1e9 * (M / idx.size) / max(t_read, 1e-9), t_dec, M / t_dec, float(M) * float(ulen[0]) / 1e9 / t_dec), flush=True)
        # check a sample against zlib
        idx = np.arange(0, M, max(1, M // 16))[:16]
        comp, off, clen, ulen = f.read_compressed(idx)
        res = d.decode(comp, off, clen, ulen, fetch_raw=True)
        ok = all(zlib.decompress(comp[off[k]:off[k] + clen[k]].tobytes()) == res["raw"][k, :ulen[k]].tobytes() for k in range(idx.size))
        print("zlib check of %d blocks:" % idx.size, "ok" if ok else "MISMATCH", "| compressed bytes per variant %.0f" % clen.mean())
    os.remove(path)


if __name__ == "__main__":
    main()
