"""bench.py's host-side helpers (no GPU): the I/O context of the from-files record and the oracle leg of `--bt --oracle-check`."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("rg_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_host_io_context_reports_residency_and_read_rates(tmp_path):
    b = _bench()
    path = str(tmp_path / "f.bin")
    with open(path, "wb") as fh:
        fh.write(os.urandom(48 << 20))
    r = b.host_io_context(path)
    assert r["page_cache_resident_fraction"] is not None and 0.0 <= r["page_cache_resident_fraction"] <= 1.0
    assert set(r["pread_GBps_by_threads"]) == {"4", "8", "16"} and all(v is None or v > 0 for v in r["pread_GBps_by_threads"].values())
    assert r["sweep_bytes"] == 48 << 20 and r["host"]["hardware_threads"] >= 1
    assert b.host_io_context(path, sweep_bytes=0)["pread_GBps_by_threads"] == {"4": None, "8": None, "16": None}


def test_bt_oracle_leg_is_zero_on_the_oracles_own_answer():
    """The comparison record of `bench.py --bt --oracle-check`: fed the oracle's own coefficients and fold sums it reports zero differences; a
    perturbed coefficient shows up at its size."""
    b = _bench()
    from oracle import regenie_step1 as orc
    rng = np.random.default_rng(3)
    N, L = 600, 12
    W = rng.standard_normal((N, L))
    liab = W[:, :4].sum(axis=1) * 0.4 + rng.standard_normal(N)
    y = (liab > np.quantile(liab, 0.8)).astype(np.float64)
    mask = np.ones(N, bool)
    off = np.full(N, np.log(0.2 / 0.8))
    cv = orc.set_folds(mask, 5)
    tau = orc.tau_from_h(orc.set_ridge_params(3), L, True)
    opt = orc.Step1Options(bed="", pheno_file="", bt=True)
    cs, betas, ok = orc.ridge_logistic_level_1(W, y, off, mask, cv, tau, opt, folds=[0])
    assert ok
    rec = b.bt_oracle_leg(W, y, off, mask, cv, tau, 0, betas[0].T.copy(), cs, float(y.mean()), False)
    assert rec["oracle_converged"] and rec["beta_max_rel_err"] == 0.0 and rec["held_out_deviance_max_rel_err"] == 0.0 and rec["prediction_max_rel_err"] == 0.0
    g = betas[0].T.copy()
    g[1, 3] *= 1.0 + 1e-3
    rec = b.bt_oracle_leg(W, y, off, mask, cv, tau, 0, g, cs, float(y.mean()), True)
    assert 0.0 < rec["beta_max_rel_err"] < 2e-3 and "quasi-Newton" in rec["route"]


def test_traffic_files_are_keyed_on_the_kernel_library_not_on_the_host_driver(tmp_path, monkeypatch):
    """build.library_digest() covers what librg_step1_hip.so is built from and nothing of regenie_amd/host; measured_traffic() takes a file
    whose library stamp is the current one even when the full build stamp differs, and refuses one measured on other kernel sources."""
    import json
    from regenie_amd import build
    lib_deps, all_deps = set(map(os.path.realpath, build._lib_deps())), set(map(os.path.realpath, build._deps()))
    assert lib_deps < all_deps and all(os.sep + "host" + os.sep in d for d in all_deps - lib_deps)
    assert not any(os.sep + "host" + os.sep in d for d in lib_deps)
    assert any(d.endswith("chol.hip") for d in lib_deps) and build.library_digest() != build.source_digest()
    b = _bench()
    root = tmp_path / "r"
    (root / "regenie_amd" / "lib").mkdir(parents=True)
    (root / "profiles").mkdir()
    (root / "regenie_amd" / "lib" / "build.stamp").write_text("build-now")
    (root / "regenie_amd" / "lib" / "library.stamp").write_text("lib-now")
    monkeypatch.setattr(b, "ROOT", str(root))
    rec = {"build_stamp": "build-then", "library_stamp": "lib-now", "level0_batches": 2, "blocks": 109, "phenos": 1,
           "groups": {"chol": {"hbm_bytes": 50.0, "lead_launches": 2, "group_launches": 8}}}
    (root / "profiles" / "x_traffic.json").write_text(json.dumps(rec))
    per, note = b.measured_traffic("chol_f64", 109, 2, 1)
    assert per == 25.0 and "x_traffic.json" in note
    rec["library_stamp"] = "lib-then"
    (root / "profiles" / "x_traffic.json").write_text(json.dumps(rec))
    per, note = b.measured_traffic("chol_f64", 109, 2, 1)
    assert per is None and "stale" in note
    # another file of the library changed (the BGEN decoder): the Cholesky group's sources are what they were, its traffic stands; a change to
    # chol.hip itself makes it stale
    files = {"csrc/rg_api.hip": "a", "csrc/rg_internal.h": "b", "csrc/bed_prep.hip": "c", "flags": "f", "csrc/chol.hip": "d", "csrc/assemble.hip": "e",
             "csrc/chol_p128.h": "g", "csrc/chol_common.h": "h", "csrc/bgen_inflate.hip": "old"}
    rec["source_digests"] = files
    (root / "profiles" / "x_traffic.json").write_text(json.dumps(rec))
    (root / "regenie_amd" / "lib" / "kernel_files.json").write_text(json.dumps(dict(files, **{"csrc/bgen_inflate.hip": "new"})))
    per, note = b.measured_traffic("chol_f64", 109, 2, 1)
    assert per == 25.0
    (root / "regenie_amd" / "lib" / "kernel_files.json").write_text(json.dumps(dict(files, **{"csrc/chol.hip": "new"})))
    per, note = b.measured_traffic("chol_f64", 109, 2, 1)
    assert per is None and "stale" in note
    (root / "regenie_amd" / "lib" / "kernel_files.json").write_text(json.dumps(dict(files, **{"csrc/chol_p128.h": "new"})))      # the panel-128 kernels (round 6)
    per, note = b.measured_traffic("chol_f64", 109, 2, 1)
    assert per is None and "stale" in note
    assert set(build.kernel_file_digests()) >= {"csrc/chol.hip", "csrc/chol_p128.h", "csrc/chol_common.h", "csrc/rg_api.hip", "csrc/rg_internal.h", "csrc/bed_prep.hip", "csrc/assemble.hip", "csrc/l1.hip", "csrc/l1x.hip",
                                                "csrc/gram_fp4.hip", "csrc/pred.hip", "csrc/pred_i8.hip", "csrc/wgram_bf16.hip", "flags"}


def test_summary_digest_of_a_recorded_line_fits_the_tail_a_reader_keeps():
    """bench.py ends its JSON line with `summary` (every sub-run's headline figures): the driver's record keeps the last 2,000 characters of the
    line, so the digest must stay well below that whatever the sub-records hold, and must carry the figures the review reads first."""
    import glob
    import json
    b = _bench()
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[56]_bench_line.json"))):
        line = json.load(open(fn))
        s = b.summary_of(line)
        assert len(json.dumps(s)) < 1500, fn
        assert s["cfg1"]["ms"] == pytest.approx(line["ms_per_step"], rel=1e-3) and s["cfg1"]["chol_frac"] == pytest.approx(line["roofline"]["frac"], rel=1e-3)
        assert s["cfg2_1gpu"]["files_s"] is not None and s["cfg2_1gpu"]["loco_ck"] == line["config3_single_gpu"]["loco_checksum"]
        assert s["cfg3_bt"]["s_per_trait"] is not None and s["step2"]["bgen_lines_identical"].count("/") == 1
    # a line whose sub-runs failed still gets a digest, with the failures named
    s = b.summary_of({"ms_per_step": 30.0, "roofline": {"frac": 0.4}, "config3_single_gpu": {"error": "x"}, "step2": {"error": "y"}})
    assert s["errors_in"] == ["config3_single_gpu", "step2"] and s["cfg2_1gpu"]["ms"] is None
