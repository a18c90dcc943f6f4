"""Pins the oracle (CPU, no GPU):
 (1) the reference's only Step-1 known answer -- `0.4504` on the `min value` line of the BT/auto-LOOCV
     run of test/test_bash.sh:62-89;
 (2) the reference's split-l0 == single-run identity (test/test_bash.sh:91-138), on the .loco text;
 (3) the extended tables two independent restatements agree on.
"""
import os

import numpy as np
import pytest

from oracle import regenie_step1 as orc


def _bt_opt(E, **kw):
    return orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype_bin.txt"),
                            covar_file=os.path.join(E, "covariates.txt"),
                            remove=[os.path.join(E, "fid_iid_to_remove.txt")],
                            exclude=[os.path.join(E, "snplist_rm.txt")], bsize=100, bt=True, **kw)


@pytest.fixture(scope="module")
def bt_run(example_dir):
    return orc.run_step1(_bt_opt(example_dir))


def test_reference_known_answer_0p4504(bt_run):
    lines = [l for l in bt_run.log if "min value" in l]
    assert any("0.4504" in l for l in lines), lines          # grep "0.4504" | grep "min value"
    assert bt_run.use_loocv                                   # N=494 < 5000 -> LOOCV (Data.cpp:353)
    assert bt_run.prep.Y.shape == (494, 2) and len(bt_run.blocks) == 10
    assert sum(b[2] for b in bt_run.blocks) == 994


def test_bt_loocv_table_matches_independent_restatement(bt_run):
    exp = {0: [(0.369652, 0.115956, 0.356128), (0.374061, 0.115212, 0.356969), (0.38248, 0.11366, 0.354553),
               (0.394005, 0.111543, 0.351498), (0.406093, 0.109482, 0.351694)],
           1: [(0.441966, 0.0990758, 0.320462), (0.443396, 0.098825, 0.318952), (0.446547, 0.0982934, 0.317855),
               (0.45046, 0.0976565, 0.317055), (0.450534, 0.0980023, 0.320258)]}
    for ph in (0, 1):
        cs, neff = bt_run.cumsum[ph], bt_run.prep.Neff[ph]
        for j, (rsq, mse, ll) in enumerate(exp[ph]):
            num = cs[4, j] - cs[0, j] * cs[1, j] / neff
            r = num * num / ((cs[2, j] - cs[0, j] ** 2 / neff) * (cs[3, j] - cs[1, j] ** 2 / neff))
            assert r == pytest.approx(rsq, rel=6e-6)
            assert (cs[2, j] + cs[3, j] - 2 * cs[4, j]) / neff == pytest.approx(mse, rel=6e-6)
            assert cs[5, j] / neff == pytest.approx(ll, rel=6e-6)
        assert bt_run.best[ph] == 3


def test_qt_kfold_config1_tables(example_dir):
    E = example_dir
    r = orc.run_step1(orc.Step1Options(bed=os.path.join(E, "example"), pheno_file=os.path.join(E, "phenotype.txt"),
                                       covar_file=os.path.join(E, "covariates.txt"), bsize=100))
    mse = [[0.980679, 0.974989, 0.976508, 0.981858, 0.99913], [0.979327, 0.985807, 0.99473, 1.00507, 1.02064]]
    for ph in (0, 1):
        cs = r.cumsum[ph]
        got = (cs[2] + cs[3] - 2 * cs[4]) / r.prep.Neff[ph]
        assert np.allclose(got, mse[ph], rtol=6e-6)
    assert list(r.best) == [1, 0]
    # single chromosome: chr-1 LOCO is exactly 0, the other 22 rows are the full PRS (Data.cpp:1846-1858)
    assert np.all(r.loco[0][:, 0] == 0)
    assert np.allclose(r.loco[0][:5, 1], [0.0985587548, 0.0851920225, -0.1926537262, -0.1092198842, 0.0678073814], atol=1e-9)
    assert np.allclose(r.loco[1][:5, 5], [0.0125497299, 0.0643410133, 0.0049116303, -0.0969964973, 0.0593433179], atol=1e-9)


def test_qt_kfold_3chr_loco(example_dir):
    E = example_dir
    r = orc.run_step1(orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                                       covar_file=os.path.join(E, "covariates.txt"), bsize=100))
    assert len(r.blocks) == 6 and list(r.best) == [1, 1]
    assert np.allclose(r.loco[0][:3, :3].T, [[-0.2066951705, -0.1673228499, 0.1541021496],
                                              [-0.0550671954, -0.0138440003, 0.019169898],
                                              [-0.1761480818, -0.0861407485, 0.12147054]], atol=1e-9)
    assert np.allclose(r.loco[1][:3, :3].T, [[0.0833061255, -0.0660699344, 0.2000572348],
                                              [0.0675113639, -0.0397203572, 0.013329815],
                                              [0.1595366331, -0.1008284436, 0.0273957686]], atol=1e-9)


def test_split_l0_equals_single_run(example_dir, tmp_path, bt_run):
    """test/test_bash.sh:91-138: level 0 split in 4 jobs (global M for lambda, Data.cpp:607), level 1 on the
    concatenated predictors -> byte-identical .loco files."""
    opt = _bt_opt(example_dir, out=str(tmp_path / "single"))
    single = orc.run_step1(opt, write_files=True)
    bim, chrom, offs, snp_ids, prep = orc.load_inputs(opt)
    blocks = orc.chrom_blocks(chrom, bim.chr_read, opt.bsize)
    B, njobs = len(blocks), 4
    bed, _ = orc.open_bed(opt.bed + ".bed", prep.n_file)
    lam = chrom.size * (1 - orc.set_ridge_params(5)) / orc.set_ridge_params(5)
    N, P = prep.Y.shape
    W = [np.zeros((N, B * 5)) for _ in range(P)]
    # job split of write_l0_master (Data.cpp:270-302): floor(B/n) blocks each, first B mod n get one more
    nb = [B // njobs + (1 if j < B % njobs else 0) for j in range(njobs)]
    b0 = 0
    for j in range(njobs):
        for b in range(b0, b0 + nb[j]):
            _, s, bs = blocks[b]
            G = orc.read_chunk_from_bed(np.asarray(bed[offs[s:s + bs]]), prep.n_file, prep.ind_ignore, prep.ind_in_analysis)
            G, _ = orc.residualize_genotypes(G, prep)
            Wb = orc.ridge_level_0_loocv(G, prep, lam)
            for ph in range(P):
                W[ph][:, b * 5:(b + 1) * 5] = Wb[ph]
        b0 += nb[j]
    opt2 = _bt_opt(example_dir, out=str(tmp_path / "split"))
    orc.finish_level_1(opt2, prep, blocks, bim.chr_read, None, lam, orc.set_ridge_params(5), W, True, [], write_files=True)
    for k in (1, 2):
        a = open(str(tmp_path / ("single_%d.loco" % k)), "rb").read()
        b = open(str(tmp_path / ("split_%d.loco" % k)), "rb").read()
        assert a == b and len(a) > 1000
    assert single.log == bt_run.log


def test_loco_file_format(example_dir, tmp_path):
    E = example_dir
    opt = orc.Step1Options(bed=os.path.join(E, "example_3chr"), pheno_file=os.path.join(E, "phenotype.txt"),
                           covar_file=os.path.join(E, "covariates.txt"), bsize=100, out=str(tmp_path / "fmt"))
    r = orc.run_step1(opt, write_files=True)
    lines = open(str(tmp_path / "fmt_1.loco")).read().split("\n")
    assert len(lines) == 25 and lines[-1] == ""                     # header + 23 chromosomes
    hdr = lines[0].split(" ")
    assert hdr[0] == "FID_IID" and hdr[-1] == "" and len(hdr) == 502
    assert hdr[1:5] == ["100_100", "101_101", "102_102", "103_103"]  # std::map (lexicographic) order
    assert [l.split(" ")[0] for l in lines[1:24]] == [str(c) for c in range(1, 24)]
    pl = open(str(tmp_path / "fmt_pred.list")).read().split("\n")
    assert pl[0] == "Y1 " + os.path.abspath(str(tmp_path / "fmt_1.loco"))
    i = r.prep.ids.index("100_100")
    assert lines[1].split(" ")[1] == "%.6g" % r.loco[0][i, 0]
