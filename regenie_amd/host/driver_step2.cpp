// regenie-amd, the C++ host driver (see driver.h): `--step 2`.
#include "driver.h"
#include <functional>

namespace rgdrv {

// The per-variant corrections -- fit_firth_logistic_snp_fast (Step2_Models.cpp:1158-1253) and run_SPA_test_snp (:2072-2297) -- run on the
// device behind the C ABI (rg_s2_bt_correct, regenie_amd/csrc/step2_bt.hip).

// One part of a `--step 2` run: the blocks [blk_lo, blk_hi) of the run's block list (chromosomes in file order, ceil(n_chr / bsize) blocks
// each) on one device.  A run on G GPUs is G parts on G host threads -- the blocks are independent, there is no exchange -- whose result
// lines go to part files that are concatenated in block order afterwards (run_step2_all).
struct S2Part {
  int part = 0, nparts = 1, device = 0;
  int blk_lo = 0, blk_hi = INT_MAX;
  int64_t n_ignored_snps = 0, n_ignored_tests = 0;     // out
  std::vector<std::string> firth_body;                 // out: --write-null-firth lines per trait
  std::vector<std::string> files;                      // out: the part's result files, one per trait
};

int run_step2(Run& r, std::chrono::steady_clock::time_point t_start, S2Part& part) {
  const Params& p = r.p;
  const int64_t N = r.N;
  const int P = r.P, C = r.C;
  std::vector<int64_t> an;                      // analysed samples (rows handed to the device), file order
  for (int64_t i = 0; i < N; ++i) if (r.ain[i]) an.push_back(i);
  const int64_t n = (int64_t)an.size();
  bool any_missing = false;                     // filters->has_missing: a sample masked for at least one trait
  std::vector<uint8_t> has_missing(n, 0);
  for (int64_t k = 0; k < n; ++k)
    for (int q = 0; q < P; ++q)
      if (!r.mask[(size_t)q * N + an[k]]) { has_missing[k] = 1; any_missing = true; }
  const bool dense_route = getenv("RG_S2_DENSE") != nullptr;     // the fp64 route of the library (rg_s2_qt_block), kept for comparison
  const bool glm = p.bt || p.ct;                                 // binary / count traits: the score test of a generalised linear null model
  // compact, sample-fastest copies for the C ABI
  std::vector<double> Xc((size_t)C * n), Yc((size_t)P * n), resc((size_t)P * n), scf(P);
  std::vector<uint8_t> Mc((size_t)P * n);
  for (int c = 0; c < C; ++c) for (int64_t k = 0; k < n; ++k) Xc[(size_t)c * n + k] = r.X[(size_t)c * N + an[k]];
  for (int q = 0; q < P; ++q)
    for (int64_t k = 0; k < n; ++k) { Yc[(size_t)q * n + k] = (glm ? r.Yraw : r.Y)[(size_t)q * N + an[k]]; Mc[(size_t)q * n + k] = r.mask[(size_t)q * N + an[k]]; }
  // binary traits (compute_res_bin, Data.cpp:2439-2445; compute_score_bt, Step2_Models.cpp:471-552): per chromosome the null logistic
  // model with the LOCO offset gives p^, w = p^ (1 - p^); the score test of a variant needs, per trait, sum w g~^2, X^T W g~ and
  // g~ . (y - p^) -- contractions of the hard-call row with fixed columns, which rg_s2_contract_packed evaluates on the i8 matrix cores
  std::vector<double> bt_fit, bt_vstat;
  std::vector<int32_t> bt_counts;
  std::vector<uint8_t> bt_pass(P, 1), test_ignored;
  const bool firth = p.bt && p.firth, spa = p.bt && p.spa, correct = firth || spa;
  const double z_thr = correct ? norm_quantile(1.0 - 0.5 * p.pthresh) : 0.0;   // sqrt of the chi-square(1) quantile at 1 - pThresh (Data.cpp:2119-2120)
  std::vector<double> firth_off;                      // [P][n] cov_blup_offset: X beta_nullFirth + LOCO prediction (fit_null_firth, Step2_Models.cpp:1011-1013)
  if (firth) firth_off.assign((size_t)P * n, 0.0);
  std::vector<double> firth_bnull((size_t)P * C, 0.0), blup_off;      // exact Firth: the covariate-only estimates (start values), the LOCO offsets
  std::vector<std::string> null_firth_files, firth_file_body(P);       // --use-null-firth: per-trait files of the list; --write-null-firth: what goes out
  if (!p.use_null_firth.empty()) {      // check_firth_file / the list reader (Step2_Models.cpp:1871-1934): `<phenotype> <file>` per line
    sout << " * reading null Firth estimates using file : [" << p.use_null_firth << "]\n";
    null_firth_files.assign(P, "");
    TextIn lf(p.use_null_firth);
    if (!lf) throw std::runtime_error("cannot read file : " + p.use_null_firth);
    std::string ln;
    while (std::getline(lf, ln)) {
      const auto t = split_ws(ln);
      if (t.empty()) continue;
      if (t.size() != 2) throw std::runtime_error("incorrectly formatted file specified by --use-null-firth.");
      for (int q = 0; q < P; ++q) if (r.pheno_names[q] == t[0]) null_firth_files[q] = t[1];
    }
  }
  if (p.write_null_firth) sout << " * writing null Firth estimates to file\n";
  if (firth && !p.firth_approx) blup_off.assign((size_t)P * n, 0.0);
  std::vector<double> denum_v;                        // per (variant, trait): the score test's denominator
  std::vector<uint8_t> corrected, corr_fail;          // per (variant, trait) of a block
  std::vector<double> corr_beta, corr_se, corr_chisq, corr_logp;
  if (glm) bt_fit.assign((size_t)P * n, 0.5);

  rg_s2_ctx* s2 = nullptr;
  if (rg_s2_create(&s2, part.device, n, C, P) != RG_S2_OK || !s2) throw std::runtime_error("no MI355X / HIP device available (rg_s2_create failed)");
  auto s2check = [&](int rc) { if (rc != RG_S2_OK) throw std::runtime_error(rg_s2_last_error(s2)); };
  enum class In { Bed, PgenHard, Dosage };
  const In in = r.dosage_mode ? In::Dosage : (r.pgen ? In::PgenHard : In::Bed);
  const bool show_info = r.dosage_mode;                 // params.dosage_mode: the INFO column
  const int flip = (in == In::Bed && p.ref_first) ? 1 : 0;   // .pgen rows always count ALT (PgenReader::Read / ReadHardcalls)
  // check_sparse_G: params.n_samples, params.prop_zero_thr (Regenie.hpp:311); the .pgen reader counts the observed zeros itself
  s2check(rg_s2_set_sparse_rule(s2, N, 0.5, r.pgen ? 1 : 0));

  // blocks per chromosome (set_blocks_for_testing: ceil(n_chr / bsize))
  std::map<int, std::vector<int64_t>> chr_snps;
  for (size_t j = 0; j < r.snp_chrom.size(); ++j) chr_snps[r.snp_chrom[j]].push_back((int64_t)j);
  // in_non_par (Geno.cpp:2419, :2251): outside the pseudo-autosomal regions of chromosome X the reference halves the males' calls in the
  // MAC (and, with the default dosage compensation off, nothing else) -- with no male in the sample file that is the autosomal rule
  if (chr_snps.count(p.nchrom) && r.has_male)
    throw std::runtime_error("--step 2 on chromosome " + std::to_string(p.nchrom) + " (X) with male samples: the sex-aware allele counts of the non-PAR region "
                             "are not built; test the autosomes (or supply a sample file without sex codes of 1).");
  int total_blocks = 0;
  for (auto& kv : chr_snps) total_blocks += (int)((kv.second.size() + p.bsize - 1) / p.bsize);
  sout << std::left << std::setw(20) << " * block size" << ": [" << p.bsize << "]\n";
  sout << std::left << std::setw(20) << " * # blocks" << ": [" << total_blocks << "]\n";
  sout << " * approximate memory usage : n/a (genotype blocks are tested on the GPU)\n";
  sout << " * using minimum MAC of " << p.min_mac << " (variants with lower MAC are ignored)\n";

  // output files, one per phenotype (split_by_pheno is the default; print_header_output_single, Step2_Models.cpp:2386-2398)
  std::vector<std::unique_ptr<TextOut>> ofs;
  std::vector<std::string> out_names;
  const bool multi = part.nparts > 1;        // parts write plain part files; run_step2_all concatenates (and compresses) them
  for (int q = 0; q < P; ++q) {
    out_names.push_back(p.out + "_" + r.pheno_names[q] + ".regenie" + (multi ? ".part" + std::to_string(part.part) : (p.gz ? ".gz" : "")));
    ofs.emplace_back(new TextOut(out_names.back(), multi ? false : p.gz));
    if (!*ofs.back()) throw std::runtime_error("cannot write file : " + out_names.back());
    if (part.part == 0) *ofs.back() << "CHROM GENPOS ID ALLELE0 ALLELE1 A1FREQ " << (show_info ? "INFO " : "") << "N TEST BETA SE CHISQ LOG10P EXTRA\n";
  }
  part.files = out_names;

  const int fd = in == In::Bed ? open((p.bed + ".bed").c_str(), O_RDONLY) : -1;
  if (in == In::Bed && fd < 0) throw std::runtime_error("cannot read bed file");
  std::vector<int64_t> file_idx(n, 0);          // file index of every analysed sample
  {
    int64_t kept = 0, k = 0;
    for (int64_t i = 0; i < r.n_file && k < n; ++i) {
      if (r.ind_ignore[i]) continue;
      if (kept == an[k]) file_idx[k++] = i;
      ++kept;
    }
  }
  int nthreads = p.threads > 0 ? p.threads : std::max(1, usable_cpus() - 1);   // Regenie.cpp:1104-1106
  nthreads = std::max(1, std::min(nthreads, 64) / part.nparts);
  // buildLookupTable (Geno.cpp:2833-2856): 00 -> 2, 01 -> missing (-3), 10 -> 1, 11 -> 0 copies of the first .bim allele
  static const double lut[4] = {2.0, -3.0, 1.0, 0.0};
  std::vector<uint8_t> rows, packed;
  std::vector<double> G, stats, bhat, sfac, mean_v, totp_v, dbuf, ibuf;
  std::vector<int32_t> ign, nobs_v, nobsp_v;
  std::vector<int64_t> vidx;
  std::vector<uint16_t> G16;
  bool identity = n == r.n_file;                 // every sample of the file is analysed, in file order
  for (int64_t k = 0; identity && k < n; ++k) identity = file_idx[k] == k;
  int64_t n_ignored_snps = 0, n_ignored_tests = 0, n_tested = 0;
  int block = 0;
  // .bed rows of a block: runs of consecutive variants are cut into pieces read by several threads (the page-cache copy of one pread is a
  // single core's memcpy), and the NEXT block of the chromosome is read while the current one is tested
  std::vector<uint8_t> rows_ahead;
  std::future<void> ahead;
  auto read_bed = [&](const std::vector<int64_t>& snps, int64_t j0, int bs, std::vector<uint8_t>& buf) {
    buf.resize((size_t)bs * r.bpr);
    struct Piece { int64_t file_off, buf_off, len; };
    std::vector<Piece> pieces;
    const int64_t chunk = 16 << 20;
    for (int j = 0; j < bs;) {
      int e = j + 1;
      while (e < bs && r.snp_offset[snps[j0 + e]] == r.snp_offset[snps[j0 + e - 1]] + 1) ++e;
      const int64_t want = (int64_t)(e - j) * r.bpr, off = 3 + r.snp_offset[snps[j0 + j]] * r.bpr;
      for (int64_t o = 0; o < want; o += chunk) pieces.push_back({off + o, (int64_t)j * r.bpr + o, std::min(chunk, want - o)});
      j = e;
    }
    std::atomic<int> failed(0);
    parallel_for((int)pieces.size(), std::min(nthreads, 8), [&](int t) {
      int64_t got = 0;
      while (got < pieces[t].len) {
        const ssize_t k = pread(fd, buf.data() + pieces[t].buf_off + got, (size_t)(pieces[t].len - got), pieces[t].file_off + got);
        if (k <= 0) { failed = 1; return; }
        got += k;
      }
    });
    if (failed) throw std::runtime_error("cannot read bed file");
  };
  // 8-bit .bgen blocks (the UK Biobank encoding): the host threads inflate the NEXT block of this part and walk its bytes once -- 2-byte
  // integer dosages (units of 1 / 255) into a pinned buffer, the allele / info sums of parseSnpfromBGEN (Geno.cpp:2186-2330) in the
  // reference's order -- while the device tests the current block and its result lines are formatted.  The three double rows per variant
  // of the general route (dosage, info term, analysed-sample copy: 12 MB per variant at 500,000 samples) do not exist on this one.
  struct DosPrep {
    uint16_t* g16 = nullptr;                 // pinned, bsize rows of ld16 entries
    std::vector<uint8_t> raw, ignored;
    std::vector<double> total, info_num, af_t, info_t;
    std::vector<int64_t> ns1, ns_t;
    bool integral = false;
    double ms_inflate = 0, ms_walk = 0, ms_wall = 0;
    std::string err;
    // device route (csrc/bgen_inflate.hip): the stored zlib streams in page-locked memory, the dosage rows left in device memory
    int64_t g16_rows = 0;                    // rows the pinned buffer holds (the host route's; allocated when that route is first taken)
    uint8_t* comp = nullptr; int64_t comp_cap = 0;
    const uint16_t* g16_dev = nullptr; int64_t ld_dev = 0;
    int dev_rows = 0;                        // rows [0, dev_rows) of the group are in device memory (g16_dev), the others in g16 from row host_row0 on
    int host_row0 = 0;
    bool dev_bad = false;
    double ms_read = 0, ms_dev = 0;
    // the stored streams of the group that will be decoded into this slot NEXT, read while the other slot's group is decoded
    std::vector<int64_t> rd_off; std::vector<int32_t> rd_clen, rd_ulen;
    std::future<bool> rd; int64_t rd_group = -1; double rd_ms = 0;
  };
  const int64_t ld16 = (n + 7) / 8 * 8;
  // host threads of the read-ahead: inflate is the bound of this input (about 10 ms per 1.5 MB block and thread with zlib), so it takes
  // --threads as given, or every hardware thread but two, shared by the parts of a multi-GPU run
  const int nt_prep = getenv("RG_S2_PREP_THREADS") ? std::max(1, atoi(getenv("RG_S2_PREP_THREADS")))
                                                    : std::max(1, std::min(p.threads > 0 ? p.threads : usable_cpus(), 256) / part.nparts);
  const bool fast_bgen = in == In::Dosage && r.bgenh && (!dense_route || glm) && !(correct && !spa && !p.firth_approx) && !getenv("RG_S2_BGEN_ROWS");
  DosPrep preps[2];
  // BGEN inflate + byte walk on the GPU (default for zlib files; RG_S2_BGEN_HOST=1 keeps the host threads' route, which also takes every
  // block the device decoder flags: a damaged stream, another encoding).  The decoder has its own stream: it works on the next block while
  // the scoring kernels of the current one run.
  rg_bgen_dev* bdev = nullptr;
  int64_t n_dev_blocks = 0, n_host_blocks = 0;
  double ms_dev_read = 0, ms_dev_decode = 0;
  struct BlkRef { const std::vector<int64_t>* snps; int64_t j0; int bs; };
  std::vector<BlkRef> my_blocks;             // this part's blocks in the order they are tested
  size_t my_next = 0;
  int64_t bgen_block_bytes = 0;
  double ms_chr = 0, ms_prep_wall = 0, ms_prep_wait = 0, ms_device = 0, ms_format = 0, ms_inflate = 0, ms_walk = 0;
  if (fast_bgen) {
    if (rg_bgen_block_bytes(r.bgenh, &bgen_block_bytes) != RG_BGEN_OK) throw std::runtime_error(rg_bgen_last_error(r.bgenh));
    bgen_block_bytes = (bgen_block_bytes + 63) / 64 * 64;
    int32_t bcomp = 0;
    rg_bgen_info(r.bgenh, nullptr, nullptr, &bcomp, nullptr);
    if (bcomp == 1 && !getenv("RG_S2_BGEN_HOST") && rg_bgen_dev_create(&bdev, part.device) == RG_BGEN_OK) {
      const bool per_trait = any_missing || glm;
      if (rg_bgen_dev_set_samples(bdev, r.n_file, n, identity ? nullptr : file_idx.data(), per_trait ? P : 0, per_trait ? Mc.data() : nullptr) != RG_BGEN_OK) {
        rg_bgen_dev_destroy(bdev);
        bdev = nullptr;
      }
    }
    int b = 0;
    for (int chrom : r.chr_read) {
      if (!chr_snps.count(chrom)) continue;
      const std::vector<int64_t>& sn = chr_snps[chrom];
      const int nbc = (int)((sn.size() + p.bsize - 1) / p.bsize);
      for (int bb = 0; bb < nbc; ++bb, ++b)
        if (b >= part.blk_lo && b < part.blk_hi)
          my_blocks.push_back({&sn, (int64_t)bb * p.bsize, (int)std::min<int64_t>(p.bsize, (int64_t)sn.size() - (int64_t)bb * p.bsize)});
    }
  }
  // The device decoder works on one stream per wavefront and needs thousands of them in flight: the blocks of a chromosome are prepared in
  // groups of >= RG_S2_BGEN_GROUP variants (default 3,072 = the streams the GPU holds at once) whatever --bsize is; the host route keeps
  // one block per group.  A group is a range of the chromosome's variants, i.e. a BlkRef of its own.
  struct Group { BlkRef ref; size_t first_block; int dev_rows; std::vector<int> starts; };
  std::vector<Group> groups;
  std::vector<std::pair<size_t, int>> block_group;      // per block of my_blocks: its group, its first row there
  if (fast_bgen) {
    // RG_S2_BGEN_HOST_SHARE=f: the host threads take the share f of every group beside the device (whole blocks from the group's end; their
    // route gives the same result lines -- both are held to regenie's).  The device's part keeps its size -- a launch takes as long for 2,000
    // streams as for 3,072, every stream being a chain of its own -- and the host's blocks come on top.  Off by default: on a box that gives
    // the job 16 CPUs the workers take those from the reads, the chromosome set-ups and the uploads (36,864 variants at 500,000 samples: 4.3 s
    // without, 4.7 s with f = 0.25, 5.4 s with 0.38, 6.9 s with 0.5); it is for hosts with idle cores.  The share is fixed for a run, so that
    // the split does not depend on timing.
    double share = 0.0;
    if (bdev)
      if (const char* e = getenv("RG_S2_BGEN_HOST_SHARE")) share = std::min(0.9, std::max(0.0, atof(e)));
    const int dev_target = std::max(p.bsize, getenv("RG_S2_BGEN_GROUP") ? atoi(getenv("RG_S2_BGEN_GROUP")) : 3072);
    const int target = bdev ? (int)std::min(65536.0, dev_target / (1.0 - share)) : p.bsize;
    for (size_t b = 0; b < my_blocks.size(); ++b) {
      const BlkRef& br = my_blocks[b];
      if (!groups.empty()) {
        Group& g = groups.back();
        if (g.ref.snps == br.snps && g.ref.j0 + g.ref.bs == br.j0 && g.ref.bs + br.bs <= target) {
          block_group.push_back({groups.size() - 1, g.ref.bs});
          g.starts.push_back(g.ref.bs);
          g.ref.bs += br.bs;
          continue;
        }
      }
      groups.push_back({br, b, 0, {0}});
      block_group.push_back({groups.size() - 1, 0});
    }
    for (Group& g : groups) {
      g.dev_rows = bdev ? g.ref.bs : 0;
      if (bdev && share > 0.0) {      // the block boundary nearest to the device's share; a group of one block stays whole
        const double want = (1.0 - share) * g.ref.bs;
        int best = g.ref.bs;
        for (int st : g.starts) if (st > 0 && std::fabs(st - want) < std::fabs(best - want)) best = st;
        g.dev_rows = best;
      }
    }
  }
  static const struct T255 { double v[256]; T255() { for (int b = 0; b < 256; ++b) v[b] = b / 255.0; } } t255;   // the reader's prob = byte / 255.0
  // the device route of a block: false = not taken (no decoder, or a variant the decoder flagged: the host route then gives the reference's verdict)
  // reads the stored streams of a group into a slot's page-locked buffer (any thread; the handle is only read)
  auto read_streams = [&](const BlkRef& br, DosPrep& d, int rows) -> bool {
    auto t0 = std::chrono::steady_clock::now();
    const int bs = rows;
    std::vector<int64_t> vi(bs);
    for (int j = 0; j < bs; ++j) vi[j] = r.snp_offset[(*br.snps)[br.j0 + j]];
    int64_t need = 0;
    if (rg_bgen_compressed_bytes(r.bgenh, bs, vi.data(), &need) != RG_BGEN_OK) return false;
    if (d.comp_cap < need) {
      if (d.comp) rg_host_free(d.comp);
      d.comp_cap = need + need / 4;
      d.comp = (uint8_t*)rg_host_alloc((size_t)d.comp_cap);
      if (!d.comp) { d.comp_cap = 0; return false; }
    }
    d.rd_off.resize(bs); d.rd_clen.resize(bs); d.rd_ulen.resize(bs);
    const bool ok = rg_bgen_read_compressed(r.bgenh, bs, vi.data(), d.comp, d.comp_cap, d.rd_off.data(), d.rd_clen.data(), d.rd_ulen.data(), std::min(nt_prep, 32)) == RG_BGEN_OK;
    d.rd_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return ok;
  };
  // the first `rows` variants of a group on the device: their sums into d.total ... (sized by the caller), their dosage rows left in device memory
  auto prepare_dev = [&](const BlkRef& br, DosPrep& d, int slot, int64_t gi, int rows) -> bool {
    if (!bdev || rows < 1) return false;
    const int bs = rows;
    // this group's streams: read ahead (while the previous group was decoded), or now
    bool have = false;
    if (d.rd.valid()) { const bool ok = d.rd.get(); have = ok && d.rd_group == gi; }
    if (!have && !read_streams(br, d, rows)) return false;
    d.rd_group = -1;
    // the NEXT group's streams go into the other slot's buffer while this one is decoded (that slot's decode is long done; the main thread
    // only reads its sums and its device rows)
    if (gi >= 0 && (size_t)gi + 1 < groups.size() && groups[gi + 1].dev_rows > 0) {
      DosPrep& dn = preps[(gi + 1) & 1];
      if (dn.rd.valid()) dn.rd.wait();
      dn.rd_group = gi + 1;
      dn.rd = std::async(std::launch::async, [&, gn = gi + 1]() { return read_streams(groups[gn].ref, preps[gn & 1], groups[gn].dev_rows); });
    }
    const std::vector<int64_t>& off = d.rd_off;
    const std::vector<int32_t>&clen = d.rd_clen, &ulen = d.rd_ulen;
    std::vector<int32_t> status(bs), maxq(bs);
    auto t1 = std::chrono::steady_clock::now();
    const bool per_trait = any_missing || glm;
    std::vector<int64_t> sq(bs), si(bs), no(bs), sqt, sit, nt;
    if (per_trait) { sqt.resize((size_t)bs * P); sit.resize((size_t)bs * P); nt.resize((size_t)bs * P); }
    rg_bgen_dev_out o;
    memset(&o, 0, sizeof(o));
    o.sum_q = sq.data(); o.sum_info = si.data(); o.n_obs = no.data(); o.max_q = maxq.data(); o.status = status.data();
    if (per_trait) { o.sum_q_t = sqt.data(); o.sum_info_t = sit.data(); o.n_obs_t = nt.data(); }
    if (rg_bgen_dev_decode(bdev, slot, bs, d.comp, off[bs - 1] + clen[bs - 1], off.data(), clen.data(), ulen.data(), p.ref_first ? 1 : 0, &o) != RG_BGEN_OK) return false;
    for (int j = 0; j < bs; ++j) if (status[j] != 0) return false;
    bool bad = false;
    for (int j = 0; j < bs; ++j) {
      // the walk's exact integer sums in the units the host route accumulates as doubles: dosages in 1 / 255, info terms in 1 / 65025
      d.total[j] = (double)sq[j] / 255.0; d.info_num[j] = (double)si[j] / 65025.0; d.ns1[j] = no[j];
      if (maxq[j] > 510) bad = true;
      const double mac = std::min(d.total[j], 2.0 * d.ns1[j] - d.total[j]);
      // The sum here is the exact integer sum / 255; the host route and regenie add the samples' doubles in order.  A count that lands on
      // --minMAC to within that summation's rounding could fall on the other side of the `<` there: such a group goes to the host route as a
      // whole (its verdict is the reference's), so that the filter does not depend on which route a variant took.  (si == 0: every call is a
      // hard call, the doubles are integers and their sum is exact whatever the order -- the common case of a count that EQUALS --minMAC.)
      if (si[j] != 0 && std::fabs(mac - p.min_mac) <= 1e-9 * std::max(1.0, mac)) return false;
      if (mac < p.min_mac) d.ignored[j] = 1;      // compute_mac (Geno.cpp:3077-3108), autosomes
      if (per_trait)
        for (int q = 0; q < P; ++q) {      // the host route SUBTRACTS what the samples missing for trait q contribute
          d.af_t[(size_t)j * P + q] = -(double)sqt[(size_t)j * P + q] / 255.0;
          d.ns_t[(size_t)j * P + q] = -nt[(size_t)j * P + q];
          d.info_t[(size_t)j * P + q] = -(double)sit[(size_t)j * P + q] / 65025.0;
        }
    }
    d.dev_bad = bad;
    d.g16_dev = o.g16; d.ld_dev = o.ld16;
    auto t2 = std::chrono::steady_clock::now();
    d.ms_read = have ? 0.0 : d.rd_ms;      // what the read cost THIS group's preparation (read ahead: nothing)
    d.ms_dev = std::chrono::duration<double, std::milli>(t2 - t1).count();
    return true;
  };
  auto prepare = [&](const BlkRef& br, DosPrep& d, int slot, int64_t gi) {
    try {
      const int bs = br.bs;
      auto ta = std::chrono::steady_clock::now();
      std::vector<int64_t> vi(bs);
      for (int j = 0; j < bs; ++j) vi[j] = r.snp_offset[(*br.snps)[br.j0 + j]];
      const bool per_trait = any_missing || glm;
      d.g16_dev = nullptr; d.ms_read = d.ms_dev = 0; d.dev_rows = 0; d.host_row0 = 0; d.dev_bad = false;
      d.total.assign(bs, 0.0); d.info_num.assign(bs, 0.0); d.ns1.assign(bs, 0); d.ignored.assign(bs, 0);
      if (per_trait) { d.af_t.assign((size_t)bs * P, 0.0); d.ns_t.assign((size_t)bs * P, 0); d.info_t.assign((size_t)bs * P, 0.0); }
      std::atomic<int> bad(0);
      std::vector<std::string> werr(nt_prep);
      std::vector<double> w_inf(nt_prep, 0.0), w_walk(nt_prep, 0.0);
      const bool rf = p.ref_first;
      // rows [lo, bs) of the group on the host threads, into the pinned buffer from its first row on
      auto host_rows = [&](int lo) {
        if (lo >= bs) return;
        if (d.g16_rows < bs - lo) {
          if (d.g16) rg_host_free(d.g16);
          d.g16 = (uint16_t*)rg_host_alloc((size_t)(bs - lo) * ld16 * sizeof(uint16_t));
          d.g16_rows = d.g16 ? bs - lo : 0;
          if (!d.g16) throw std::runtime_error("cannot allocate the pinned dosage buffers");
        }
        d.host_row0 = lo;
        d.raw.resize((size_t)nt_prep * bgen_block_bytes);        // one inflated block per worker: walked while it is still in that core's cache
        std::atomic<int> next(lo);
        // (no reader lock: the read call only reads the handle, so the parts of a --gpus N run inflate at the same time)
        parallel_for(nt_prep, nt_prep, [&](int w) {
          uint8_t* blk = d.raw.data() + (size_t)w * bgen_block_bytes;
          for (int j; (j = next.fetch_add(1)) < bs;) {
            auto t0 = std::chrono::steady_clock::now();
            if (rg_bgen_read_blocks(r.bgenh, 1, &vi[j], blk, bgen_block_bytes, 1) != RG_BGEN_OK) { werr[w] = rg_bgen_last_error(r.bgenh); next = bs; return; }
            auto t1 = std::chrono::steady_clock::now();
            const uint8_t* ploidy = blk + 8;
            const uint8_t* pr = blk + 10 + r.n_file;
            uint16_t* q16 = d.g16 + (size_t)(j - lo) * ld16;
            double tot = 0.0, inf = 0.0; int64_t ns = 0;
            unsigned worst = 0;
            if (per_trait)
              for (int q = 0; q < P; ++q) { d.af_t[(size_t)j * P + q] = 0.0; d.ns_t[(size_t)j * P + q] = 0; d.info_t[(size_t)j * P + q] = 0.0; }
            for (int64_t k = 0; k < n; ++k) {
              const int64_t i = identity ? k : file_idx[k];
              if (ploidy[i] & 0x80) { q16[k] = 0xFFFFu; continue; }
              const unsigned b0 = pr[2 * i], b1 = pr[2 * i + 1];
              const double p0 = t255.v[b0], p1 = t255.v[b1];
              double v, e;
              unsigned qi;
              if (rf) {     // G = prob1 + 2 prob2, prob2 = max(1 - prob0 - prob1, 0) (Geno.cpp:2286-2290)
                const double p2 = std::max(1.0 - p0 - p1, 0.0);
                v = p1 + 2.0 * p2; e = (4.0 * p2 + p1) - v * v;
                qi = b1 + 2u * (b0 + b1 < 255u ? 255u - b0 - b1 : 0u);
              } else {
                v = p1 + 2.0 * p0; e = (4.0 * p0 + p1) - v * v;
                qi = b1 + 2u * b0;
              }
              worst = std::max(worst, qi);
              q16[k] = (uint16_t)qi;
              tot += v; inf += e; ++ns;
              if (per_trait && has_missing[k])
                for (int q = 0; q < P; ++q)
                  if (!Mc[(size_t)q * n + k]) { d.af_t[(size_t)j * P + q] -= v; d.ns_t[(size_t)j * P + q] -= 1; d.info_t[(size_t)j * P + q] -= e; }
            }
            for (int64_t k = n; k < ld16; ++k) q16[k] = 0;
            if (worst > 510u) bad = 1;          // prob0 + prob1 > 1 in the file: not a dosage in [0, 2], the general route reports what the reference would
            d.total[j] = tot; d.ns1[j] = ns; d.info_num[j] = inf;
            d.ignored[j] = std::min(tot, 2.0 * ns - tot) < p.min_mac ? 1 : 0;      // compute_mac (Geno.cpp:3077-3108), autosomes
            auto t2 = std::chrono::steady_clock::now();
            w_inf[w] += std::chrono::duration<double, std::milli>(t1 - t0).count();
            w_walk[w] += std::chrono::duration<double, std::milli>(t2 - t1).count();
          }
        });
        for (const auto& e : werr) if (!e.empty()) throw std::runtime_error(e);
      };
      // the group's first rows on the device and, beside them, its last blocks on the host threads; a group the decoder turns down goes to
      // the host threads as a whole (a damaged stream, another encoding: their messages are the reference's)
      const int split = (bdev && gi >= 0) ? groups[gi].dev_rows : 0;
      bool dev_ok = false;
      if (split > 0) {
        std::future<bool> fdev = std::async(std::launch::async, [&]() { return prepare_dev(br, d, slot, gi, split); });
        try { host_rows(split); } catch (...) { fdev.wait(); throw; }
        dev_ok = fdev.get();
      }
      if (dev_ok) d.dev_rows = split;
      else { d.g16_dev = nullptr; d.dev_bad = false; host_rows(0); }
      d.integral = !bad && !d.dev_bad;
      // thread-milliseconds of the two halves, and the wall time of the block's preparation
      d.ms_inflate = 0; d.ms_walk = 0;
      for (int w = 0; w < nt_prep; ++w) { d.ms_inflate += w_inf[w]; d.ms_walk += w_walk[w]; }
      d.ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();
    } catch (const std::exception& e) { d.err = e.what(); if (d.err.empty()) d.err = "bgen read failed"; }
  };
  std::future<void> prep_ahead;      // declared after everything `prepare` touches: its destructor waits for the worker before those go away
  // The read-ahead futures of the device route live in `preps` (declared BEFORE `groups`, which their worker dereferences) and the pinned
  // buffers / the device decoder are freed by hand at the end of the normal path: on an exception (a failed check, a rethrown reader error)
  // this guard -- declared after all of them, so destroyed first -- joins the workers and releases what they use, in that order.
  struct S2Cleanup {
    std::function<void()> f;
    bool done = false;
    rg_bgen_dev** dev = nullptr;
    void run() { if (!done) { done = true; f(); } }
    ~S2Cleanup() { run(); if (dev && *dev) { rg_bgen_dev_destroy(*dev); *dev = nullptr; } }
  } cleanup{[&]() {
    if (prep_ahead.valid()) prep_ahead.wait();
    for (auto& d : preps) if (d.rd.valid()) d.rd.wait();
    for (auto& d : preps) { if (d.g16) rg_host_free(d.g16); if (d.comp) rg_host_free(d.comp); d.g16 = nullptr; d.comp = nullptr; }
  }};
  cleanup.dev = &bdev;
  if (fast_bgen && !groups.empty())      // the first group is inflated while the first chromosome's predictions are read
    prep_ahead = std::async(std::launch::async, [&]() { prepare(groups[0].ref, preps[0], 0, 0); });
  for (int chrom : r.chr_read) {
    if (!chr_snps.count(chrom)) continue;
    const std::vector<int64_t>& snps = chr_snps[chrom];
    const int nb_chr = (int)((snps.size() + p.bsize - 1) / p.bsize);
    if (block + nb_chr <= part.blk_lo || block >= part.blk_hi) { block += nb_chr; continue; }     // none of the chromosome's blocks is this part's
    sout << "Chromosome " << chrom << " [" << nb_chr << " blocks in total]\n";
    // blup_read_chr (Step2_Models.cpp:51-140) + compute_res (Data.cpp:2386-2400)
    sout << (p.bt ? "   -reading loco predictions for the chromosome and fitting null logistic regression on binary phenotypes..."
                  : p.ct ? "   -reading loco predictions for the chromosome and fitting null poisson regression..." : "   -reading loco predictions for the chromosome...");
    auto tb = std::chrono::steady_clock::now();
    // the chromosome's row of every phenotype's .loco file (500,000 numbers each at UK Biobank size): read and converted by one host thread
    // per phenotype, the checks reported in phenotype order
    std::vector<std::vector<double>> blup_q(P);
    std::vector<std::string> blup_err(P);
    parallel_for(P, nthreads, [&](int q) {
      Run::Blup& bl = r.blups[q];
      if (chrom < 1 || chrom > (int)bl.line_off.size()) { blup_err[q] = "blup file for phenotype '" + r.pheno_names[q] + "' has no line for chromosome " + std::to_string(chrom) + "."; return; }
      std::string line;
      if (!bl.lines.empty()) line = bl.lines[chrom - 1];
      else {
        std::ifstream f(bl.file, std::ios::binary);
        f.seekg(bl.line_off[chrom - 1]);
        std::getline(f, line);
      }
      std::vector<Tok> t(bl.col_sample.size() + 1);
      const int nt = tokenize(line.data(), line.data() + line.size(), t.data(), (int)t.size());
      if ((size_t)nt != bl.col_sample.size()) {
        blup_err[q] = "blup file for phenotype '" + r.pheno_names[q] + "' has different number of entries on line " + std::to_string(chrom + 1) + " compared to the header (=" + std::to_string(nt) + " vs " + std::to_string(bl.col_sample.size()) + ").";
        return;
      }
      if (chr_str_to_int(std::string(t[0].b, t[0].e), p.nchrom) != chrom) {
        blup_err[q] = "blup file for phenotype '" + r.pheno_names[q] + "' starts with `" + std::string(t[0].b, t[0].e) + "`instead of chromosome number=" + std::to_string(chrom) + ".";
        return;
      }
      std::vector<double>& blup = blup_q[q];
      blup.assign(N, 0.0);
      for (int c = 1; c < nt; ++c) {
        const int64_t i = bl.col_sample[c];
        if (i < 0 || !r.ain[i] || !r.mask[(size_t)q * N + i]) continue;
        const double v = convert_double_tok(t[c].b, t[c].e);
        if (v == MISSING) { blup_err[q] = "individual has missing predictions (chr=" + std::to_string(chrom) + ";phenotype=" + r.pheno_names[q] + ")."; return; }
        blup[i] = v;
      }
    });
    for (int q = 0; q < P; ++q)
      if (!blup_err[q].empty()) throw std::runtime_error(blup_err[q]);
    const auto tb1 = std::chrono::steady_clock::now();
    for (int q = 0; q < P; ++q) {
      const std::vector<double>& blup = blup_q[q];
      if (glm) {   // fit_null_logistic / fit_null_poisson, test-mode branch (Step1_Models.cpp:54-140, :225-288): offset = the LOCO prediction of
                   // the analysed, unmasked samples
        std::vector<double> off(n), eta, pv;
        for (int64_t k = 0; k < n; ++k) off[k] = blup[an[k]] * Mc[(size_t)q * n + k];
        const double* yq = Yc.data() + (size_t)q * n;
        const uint8_t* mq = Mc.data() + (size_t)q * n;
        bool ok;
        std::vector<double> bnull;
        if (p.ct) ok = fit_poisson(yq, Xc.data(), mq, n, C, p, eta, off.data(), &pv);
        else {
          LogisticState lst;
          ok = fit_logistic(yq, Xc.data(), mq, n, C, p, true, eta, off.data(), &pv, &bnull, &lst);
          if (!ok) ok = fit_logistic(yq, Xc.data(), mq, n, C, p, false, eta, off.data(), &pv, &bnull, &lst);
        }
        if (ok && firth) {   // fit_null_firth (Step2_Models.cpp:985-1060): penalised fit of the covariates, start = the unpenalised estimate
          if (!null_firth_files.empty() && !null_firth_files[q].empty()) {   // --use-null-firth: the stored estimates of this chromosome as start
            TextIn nf(null_firth_files[q]);                                   // (get_beta_start_firth, Step2_Models.cpp:1936-1981)
            if (!nf) throw std::runtime_error("cannot read file : " + null_firth_files[q]);
            std::string ln;
            while (std::getline(nf, ln)) {
              const auto t = split_ws(ln);
              if (t.empty()) throw std::runtime_error("error reading null firth estimates file");
              if (chr_str_to_int(t[0], p.nchrom) != chrom) continue;
              if ((int)t.size() - 1 > C) throw std::runtime_error("file has more predictors than included in analysis (=" + std::to_string(t.size()) + " vs " + std::to_string(C) + ")");
              for (size_t c = 1; c < t.size(); ++c) {
                const double v = convert_double(t[c]);
                if (v == MISSING) throw std::runtime_error("no missing values allowed in file");
                bnull[c - 1] = v;
              }
              break;
            }
          }
          ok = firth_null_fit(yq, Xc.data(), mq, off.data(), n, C, bnull);
          if (ok && p.write_null_firth) {     // (*firth_est_files[i]) << chrom << " " << bvec (Step2_Models.cpp:1019-1020)
            std::ostringstream ln;
            ln << chrom << " ";
            for (int c = 0; c < C; ++c) ln << bnull[c] << (c + 1 < C ? " " : "");
            firth_file_body[q] += ln.str() + "\n";
          }
          if (!ok) sout << "\n     WARNING: null Firth failed for phenotype '" << r.pheno_names[q] << "' (it will be skipped).";
          for (int64_t k = 0; ok && k < n; ++k) {
            double e = blup[an[k]];
            for (int c = 0; c < C; ++c) e += Xc[(size_t)c * n + k] * bnull[c];
            firth_off[(size_t)q * n + k] = e;
            if (!p.firth_approx) blup_off[(size_t)q * n + k] = blup[an[k]];
          }
          for (int c = 0; ok && c < C; ++c) firth_bnull[(size_t)q * C + c] = bnull[c];
        }
        bt_pass[q] = ok ? 1 : 0;
        if (!ok) { if (!(firth && !bnull.empty())) sout << "\n     WARNING: " << (p.ct ? "poisson" : "logistic") << " regression did not converge for phenotype '" << r.pheno_names[q] << "'."; continue; }
        // the fitted mean of the null model: the library forms Gamma_sqrt^2, the weighted covariates and (X^T W X)^-1 from it (rg_s2_bt_set_null)
        for (int64_t k = 0; k < n; ++k) bt_fit[(size_t)q * n + k] = pv[k];
        continue;
      }
      double ss = 0.0;
      for (int64_t k = 0; k < n; ++k) {
        const double v = (Yc[(size_t)q * n + k] - blup[an[k]]) * Mc[(size_t)q * n + k];
        resc[(size_t)q * n + k] = v;
        ss += v * v;
      }
      const double sd = std::sqrt(ss) / std::sqrt(r.neff[q] - C);
      for (int64_t k = 0; k < n; ++k) resc[(size_t)q * n + k] /= sd;
      scf[q] = r.scale_Y[q] * sd;
    }
    const auto tb2 = std::chrono::steady_clock::now();
    if (glm) {   // compute_res_bin / compute_res_count (Data.cpp:2439-2455): the null models of the chromosome go to the device
      rg_s2_bt_null nm;
      memset(&nm, 0, sizeof(nm));
      nm.family = p.ct ? 1 : 0; nm.niter_max = p.niter_max; nm.X = Xc.data(); nm.y = Yc.data(); nm.mask = Mc.data(); nm.fitted = bt_fit.data();
      nm.firth_offset = (firth && p.firth_approx) ? firth_off.data() : nullptr; nm.pass = bt_pass.data();
      s2check(rg_s2_bt_set_null(s2, &nm));
    } else s2check(rg_s2_set_null(s2, Xc.data(), resc.data(), Mc.data(), scf.data()));
    sout << "done (" << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - tb).count() << "ms) \n";
    ms_chr += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
    if (getenv("RG_TIMING"))
      fprintf(stderr, "[timing] chromosome %d set-up: predictions read + converted %.0f ms, null models / residuals %.0f ms, to the device %.0f ms\n", chrom,
              std::chrono::duration<double, std::milli>(tb1 - tb).count(), std::chrono::duration<double, std::milli>(tb2 - tb1).count(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb2).count());

    for (int bb = 0; bb < nb_chr; ++bb, ++block) {
      if (block < part.blk_lo || block >= part.blk_hi) continue;
      const int64_t j0 = (int64_t)bb * p.bsize;
      const int bs = (int)std::min<int64_t>(p.bsize, (int64_t)snps.size() - j0);
      sout << " block [" << block + 1 << "/" << total_blocks << "] : ";
      auto t1 = std::chrono::steady_clock::now();
      vidx.resize(bs);
      for (int j = 0; j < bs; ++j) vidx[j] = r.snp_offset[snps[j0 + j]];
      if (in != In::Dosage) rows.resize((size_t)bs * r.bpr);
      std::unique_lock<std::mutex> rlk(g_reader_mu, std::defer_lock);
      if (multi && in != In::Bed) rlk.lock();
      DosPrep* dp = nullptr;
      int dp_row0 = 0;                 // the block's first row in its prepared group
      if (fast_bgen) {
        if (rlk.owns_lock()) rlk.unlock();      // (the prepared block took the reader's lock itself)
        auto tw = std::chrono::steady_clock::now();
        const size_t gi = block_group[my_next].first;
        dp_row0 = block_group[my_next].second;
        DosPrep& d = preps[gi & 1];
        if (groups[gi].first_block == my_next) {      // first block of its group: the group has to be ready, the next one is started
          if (prep_ahead.valid()) prep_ahead.get();
          else prepare(groups[gi].ref, d, (int)(gi & 1), (int64_t)gi);
          if (gi + 1 < groups.size())
            prep_ahead = std::async(std::launch::async, [&, nx = gi + 1]() { prepare(groups[nx].ref, preps[nx & 1], (int)(nx & 1), (int64_t)nx); });
          if (!d.err.empty()) { if (prep_ahead.valid()) prep_ahead.wait(); throw std::runtime_error(d.err); }
          ms_inflate += d.ms_inflate; ms_walk += d.ms_walk; ms_prep_wall += d.ms_wall;
          if (d.dev_rows > 0) { ms_dev_read += d.ms_read; ms_dev_decode += d.ms_dev; }
        }
        ++my_next;
        ms_prep_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
        if (dp_row0 < d.dev_rows) ++n_dev_blocks; else ++n_host_blocks;
        if (d.integral) dp = &d;
      }
      if (in == In::PgenHard) {   // ReadHardcalls per variant (Geno.cpp:2570-2573), as .bed-coded rows (00 = two ALT copies)
        if (rg_pgen_read_bed_rows(r.pgen, bs, vidx.data(), rows.data(), r.bpr) != RG_PGEN_OK) throw std::runtime_error(rg_pgen_last_error(r.pgen));
      } else if (in == In::Dosage && !dp) {
        dbuf.resize((size_t)bs * r.n_file);
        if (r.bgenh) {            // parseSnpfromBGEN (Geno.cpp:2186-2330): dosages and the terms of the IMPUTE info score
          ibuf.resize((size_t)bs * r.n_file);
          if (rg_bgen_read_dosages_info(r.bgenh, bs, vidx.data(), p.ref_first ? 1 : 0, dbuf.data(), ibuf.data(), r.n_file) != RG_BGEN_OK)
            throw std::runtime_error(rg_bgen_last_error(r.bgenh));
        } else if (rg_pgen_read_dosage_rows(r.pgen, bs, vidx.data(), dbuf.data(), r.n_file) != RG_PGEN_OK)   // Read() (Geno.cpp:2570-2571)
          throw std::runtime_error(rg_pgen_last_error(r.pgen));
      }
      if (rlk.owns_lock()) rlk.unlock();
      if (in == In::Bed) {   // the block's rows: read ahead by the previous iteration when it could be (same chromosome), else read now
        if (ahead.valid()) { ahead.get(); rows.swap(rows_ahead); }
        else read_bed(snps, j0, bs, rows);
        if (bb + 1 < nb_chr && block + 1 < part.blk_hi) {
          const int64_t jn = (int64_t)(bb + 1) * p.bsize;
          const int bn = (int)std::min<int64_t>(p.bsize, (int64_t)snps.size() - jn);
          ahead = std::async(std::launch::async, [&, jn, bn]() { read_bed(snps, jn, bn, rows_ahead); });
        }
      }
      auto t_dev = std::chrono::steady_clock::now();
      std::vector<double> total(bs, 0.0);
      std::vector<int64_t> ns1(bs, 0);
      std::vector<double> af_t, mac_t, info_num, info_t;   // per trait: only filled when some sample is masked for some trait
      std::vector<int64_t> ns_t;
      std::vector<uint8_t> variant_ignored(bs, 0);
      rg_s2_qt_out o;
      stats.resize((size_t)bs * P); bhat.resize((size_t)bs * P); sfac.resize(bs); ign.resize(bs);
      memset(&o, 0, sizeof(o));
      o.stats = stats.data(); o.bhat = bhat.data(); o.scale_fac = sfac.data(); o.ignored = ign.data();
      test_ignored.assign((size_t)bs * P, 0);
      bool integral = false;
      const int dscale = r.bgenh ? 255 : 16384;
      const uint16_t* g16p = nullptr;
      int64_t g16ld = n;
      int g16_on_device = 0;
      if (dp) {      // the block the host threads prepared ahead
        const size_t r0 = (size_t)dp_row0;
        total.assign(dp->total.begin() + r0, dp->total.begin() + r0 + bs); ns1.assign(dp->ns1.begin() + r0, dp->ns1.begin() + r0 + bs);
        info_num.assign(dp->info_num.begin() + r0, dp->info_num.begin() + r0 + bs);
        variant_ignored.assign(dp->ignored.begin() + r0, dp->ignored.begin() + r0 + bs);
        if (any_missing || glm) {
          af_t.assign(dp->af_t.begin() + r0 * P, dp->af_t.begin() + (r0 + bs) * P); ns_t.assign(dp->ns_t.begin() + r0 * P, dp->ns_t.begin() + (r0 + bs) * P);
          info_t.assign(dp->info_t.begin() + r0 * P, dp->info_t.begin() + (r0 + bs) * P);
        }
        const bool on_dev = dp_row0 < dp->dev_rows;      // (a group is split at a block boundary)
        integral = true; g16ld = on_dev ? dp->ld_dev : ld16;
        g16p = on_dev ? dp->g16_dev + r0 * (size_t)g16ld : dp->g16 + (r0 - (size_t)dp->host_row0) * (size_t)g16ld;
        g16_on_device = on_dev ? 1 : 0;
      } else if (in == In::Dosage) {
        // dosages: the analysed samples' doubles, allele totals, the info-score numerator and the per-trait corrections on the host
        // (parseSnpfromBGEN / readChunkFromPGENFileToG with update_trait_counts, Geno.cpp:2948-2959), the test on the fp64 route
        G.assign((size_t)bs * n, 0.0);
        info_num.assign(bs, 0.0);
        if (any_missing || glm) { af_t.assign((size_t)bs * P, 0.0); ns_t.assign((size_t)bs * P, 0); info_t.assign((size_t)bs * P, 0.0); }
        parallel_for(bs, nthreads, [&](int j) {
          const double* d = dbuf.data() + (size_t)j * r.n_file;
          const double* iv = r.bgenh ? ibuf.data() + (size_t)j * r.n_file : nullptr;
          double* g = G.data() + (size_t)j * n;
          double tot = 0.0, inf = 0.0; int64_t ns = 0;
          for (int64_t k = 0; k < n; ++k) {
            const int64_t i = file_idx[k];
            const double v = d[i];
            g[k] = v;
            if (v == -3.0) continue;
            const double e = iv ? iv[i] : v * v;
            tot += v; inf += e; ++ns;
            if ((any_missing || glm) && has_missing[k])
              for (int q = 0; q < P; ++q)
                if (!Mc[(size_t)q * n + k]) { af_t[(size_t)j * P + q] -= v; ns_t[(size_t)j * P + q] -= 1; info_t[(size_t)j * P + q] -= e; }
          }
          total[j] = tot; ns1[j] = ns; info_num[j] = inf;
          if (std::min(tot, 2.0 * ns - tot) < p.min_mac) variant_ignored[j] = 1;      // compute_mac (Geno.cpp:3077-3108), autosomes
        });
        // 8-bit .bgen probabilities and .pgen dosages are integers in units of 1 / 255 and 1 / 16384: as uint16 rows they take the
        // integer route of the library (digit planes on the i8 matrix cores, 2 B per genotype over PCIe); anything else, or
        // RG_S2_DENSE=1, the fp64 route
        integral = !dense_route || glm;
        if (integral) {
          G16.resize((size_t)bs * n);
          std::vector<uint8_t> bad(bs, 0);
          parallel_for(bs, nthreads, [&](int j) {
            const double* g = G.data() + (size_t)j * n;
            uint16_t* q = G16.data() + (size_t)j * n;
            for (int64_t k = 0; k < n; ++k) {
              if (g[k] == -3.0) { q[k] = 0xFFFFu; continue; }
              const double v = g[k] * dscale, rv = std::nearbyint(v);
              if (std::fabs(v - rv) > 1e-6 || rv < 0 || rv > 2.0 * dscale) { bad[j] = 1; break; }
              q[k] = (uint16_t)rv;
            }
          });
          for (int j = 0; j < bs; ++j) if (bad[j]) integral = false;
          g16p = G16.data();
        }
      }
      std::vector<double> af_d; std::vector<int64_t> ns_d;
      if (glm && in == In::Dosage) { af_d = af_t; ns_d = ns_t; }
      if (glm) {
        // the score test of the block through the C ABI (rg_s2_bt_score_*: contractions on the i8 matrix cores, C x C algebra in the library):
        // hard calls as packed rows, dosages as integer rows
        bt_counts.resize((size_t)bs * 4); bt_vstat.resize((size_t)bs * 4);
        denum_v.assign((size_t)bs * P, 0.0);
        std::vector<double> mu_v(bs, 0.0), totp((size_t)bs * P, 0.0);
        std::vector<uint8_t> sparse_v(bs, 0);
        std::vector<int32_t> nobsp((size_t)bs * P, 0);
        rg_s2_bt_out bo;
        memset(&bo, 0, sizeof(bo));
        bo.stats = stats.data(); bo.bhat = bhat.data(); bo.denum = denum_v.data(); bo.test_ignored = test_ignored.data(); bo.mean = mu_v.data();
        bo.ignored = ign.data(); bo.sparse = sparse_v.data();
        const uint8_t* src = rows.data();
        int64_t ld = r.bpr;
        if (in == In::Dosage) {
          if (!integral) throw std::runtime_error("--step 2 --bt / --ct on dosages that are not integer multiples of 1/" + std::to_string(dscale) + " is not built.");
          bo.vstat = bt_vstat.data();
          s2check(rg_s2_bt_score_int(s2, g16p, g16ld, bs, g16_on_device, dscale, NUMTOL, &bo));
        } else {
          if (!identity) {
            ld = (n + 3) / 4;
            packed.assign((size_t)bs * ld, 0);
            parallel_for(bs, nthreads, [&](int j) {
              const uint8_t* row = rows.data() + (size_t)j * r.bpr;
              uint8_t* dst = packed.data() + (size_t)j * ld;
              for (int64_t k = 0; k < n; ++k) {
                const int64_t i = file_idx[k];
                dst[k >> 2] |= (uint8_t)(((row[i >> 2] >> (2 * (i & 3))) & 3) << (2 * (k & 3)));
              }
            });
            src = packed.data();
          }
          bo.counts = bt_counts.data(); bo.total_p = totp.data(); bo.n_obs_p = nobsp.data();
          s2check(rg_s2_bt_score_packed(s2, src, ld, bs, 0, flip, NUMTOL, &bo));
        }
        af_t.assign((size_t)bs * P, 0.0); ns_t.assign((size_t)bs * P, 0);
        for (int j = 0; j < bs; ++j) {
          if (in != In::Dosage) {     // (dosages: the host loop above has them, summed as the reference sums)
            const double n1 = bt_counts[(size_t)j * 4], n2 = bt_counts[(size_t)j * 4 + 1], nm = bt_counts[(size_t)j * 4 + 2];
            ns1[j] = (int64_t)((double)n - nm); total[j] = n1 + 2.0 * n2;
          }
          sfac[j] = 1.0;
          if (std::min(total[j], 2.0 * ns1[j] - total[j]) < p.min_mac) variant_ignored[j] = 1;
          for (int q = 0; q < P; ++q) {
            af_t[(size_t)j * P + q] = in != In::Dosage ? totp[(size_t)j * P + q] : af_d[(size_t)j * P + q];      // per-trait allele and sample counts
            ns_t[(size_t)j * P + q] = in != In::Dosage ? (int64_t)nobsp[(size_t)j * P + q] : ns_d[(size_t)j * P + q];
          }
        }
        if (correct) {
          // check_pval_snp (Step2_Models.cpp:1987-2029): |z| above the threshold -> run_SPA_test (--spa) or fit_firth_logistic_snp_fast on Gres / Gamma_sqrt
          // with the null Firth model's covariate effects in the offset.  The flagged (variant, trait) pairs are re-tested on the device, one
          // workgroup per pair (rg_s2_bt_correct); the exact Firth test (--firth without --approx: a C + 1 parameter fit) stays on the host threads.
          corrected.assign((size_t)bs * P, 0); corr_fail.assign((size_t)bs * P, 0);
          corr_beta.assign((size_t)bs * P, 0.0); corr_se.assign((size_t)bs * P, 0.0); corr_chisq.assign((size_t)bs * P, 0.0); corr_logp.assign((size_t)bs * P, -1.0);
          std::vector<int> todo;
          for (int j = 0; j < bs; ++j)
            for (int q = 0; q < P; ++q)
              if (!variant_ignored[j] && !ign[j] && !test_ignored[(size_t)j * P + q] && std::fabs(stats[(size_t)j * P + q]) > z_thr) todo.push_back(j * P + q);
          if (spa || p.firth_approx) {
            std::vector<int32_t> pv_(todo.size()), pt_(todo.size());
            std::vector<uint8_t> pf_(todo.size());
            for (size_t t = 0; t < todo.size(); ++t) {
              const int j = todo[t] / P, q = todo[t] % P;
              pv_[t] = j; pt_[t] = q;
              if (spa) pf_[t] = sparse_v[j];                                                            // fastSPA (Step2_Models.cpp:2087-2097)
              else {
                const double tq = total[j] + af_t[(size_t)j * P + q];
                const double nsq = (double)(ns1[j] + ns_t[(size_t)j * P + q]);
                pf_[t] = sparse_v[j] && std::min(tq, 2.0 * nsq - tq) < 50.0;                            // fit_firth_logistic_snp_fast :1173-1185: carriers only
              }
            }
            std::vector<rg_s2_bt_corr> cr(todo.size());
            s2check(rg_s2_bt_correct(s2, spa ? RG_S2_BT_SPA : RG_S2_BT_FIRTH_APPROX, (int32_t)todo.size(), pv_.data(), pt_.data(), pf_.data(), p.firth_se ? 1 : 0, cr.data()));
            for (size_t t = 0; t < todo.size(); ++t) {
              const size_t e = (size_t)todo[t];
              corrected[e] = 1;
              if (cr[t].fail) { corr_fail[e] = 1; continue; }
              corr_beta[e] = cr[t].beta; corr_se[e] = cr[t].se; corr_chisq[e] = cr[t].chisq; corr_logp[e] = cr[t].logp;
            }
          } else
          parallel_for((int)todo.size(), nthreads, [&](int t) {
            const int j = todo[t] / P, q = todo[t] % P;
            const double mu = mu_v[j];
            std::vector<double> gt(n);                // the mean-imputed genotype of the analysed samples
            if (in == In::Dosage) { const double* g = G.data() + (size_t)j * n; for (int64_t k = 0; k < n; ++k) gt[k] = g[k] == -3.0 ? mu : g[k]; }
            else {
              const uint8_t* row = src + (size_t)j * ld;
              for (int64_t k = 0; k < n; ++k) {
                double hc = lut[(row[k >> 2] >> (2 * (k & 3))) & 3];
                if (flip && hc != -3.0) hc = 2.0 - hc;
                gt[k] = hc == -3.0 ? mu : hc;
              }
            }
            // the exact test (fit_firth_logistic_snp, Step2_Models.cpp:1062-1156): design [covariates | g~ on its raw scale], offset = the LOCO
            // prediction; null fit = the variant's coefficient held at 0 under the same penalty, then every coefficient free
            const uint8_t* mq = Mc.data() + (size_t)q * n;
            std::vector<const double*> cols(C + 1);
            for (int c = 0; c < C; ++c) cols[c] = Xc.data() + (size_t)c * n;
            cols[C] = gt.data();
            std::vector<double> bf(C + 1, 0.0), inv;
            for (int c = 0; c < C; ++c) bf[c] = firth_bnull[(size_t)q * C + c];
            double dev0 = 0.0, dev1 = 0.0;
            corrected[(size_t)j * P + q] = 1;
            const bool okx = firth_fit_cols(Yc.data() + (size_t)q * n, cols, mq, blup_off.data() + (size_t)q * n, n, C, 25.0, bf, &dev0) &&
                             firth_fit_cols(Yc.data() + (size_t)q * n, cols, mq, blup_off.data() + (size_t)q * n, n, C + 1, 5.0, bf, &dev1, &inv);
            const double lrt = dev0 - dev1;
            if (!okx || lrt < 0) { corr_fail[(size_t)j * P + q] = 1; return; }
            corr_beta[(size_t)j * P + q] = bf[C];
            corr_chisq[(size_t)j * P + q] = lrt;
            corr_se[(size_t)j * P + q] = (p.firth_se && lrt > 0) ? std::fabs(bf[C]) / std::sqrt(lrt) : std::sqrt(inv[(size_t)C * (C + 1) + C]);
          });
        }
      } else if (in == In::Dosage) {
        if (integral) s2check(rg_s2_qt_block_int(s2, g16p, g16ld, bs, g16_on_device, dscale, NUMTOL, &o));
        else s2check(rg_s2_qt_block(s2, G.data(), n, bs, 0, NUMTOL, &o));
      } else if (!dense_route) {
        // hard calls stay packed: the 2-bit codes of the analysed samples go to the device as they are (the rows of the file itself
        // when no sample was dropped), the library counts the calls and contracts them on the i8 matrix cores
        const uint8_t* src = rows.data();
        int64_t ld = r.bpr;
        if (!identity) {
          ld = (n + 3) / 4;
          packed.assign((size_t)bs * ld, 0);
          parallel_for(bs, nthreads, [&](int j) {
            const uint8_t* row = rows.data() + (size_t)j * r.bpr;
            uint8_t* dst = packed.data() + (size_t)j * ld;
            for (int64_t k = 0; k < n; ++k) {
              const int64_t i = file_idx[k];
              dst[k >> 2] |= (uint8_t)(((row[i >> 2] >> (2 * (i & 3))) & 3) << (2 * (k & 3)));
            }
          });
          src = packed.data();
        }
        mean_v.resize(bs); nobs_v.resize(bs);
        o.mean = mean_v.data(); o.n_obs = nobs_v.data();
        if (any_missing) {
          totp_v.resize((size_t)bs * P); nobsp_v.resize((size_t)bs * P);
          o.total_p = totp_v.data(); o.n_obs_p = nobsp_v.data();
        }
        s2check(rg_s2_qt_block_packed(s2, src, ld, bs, 0, flip, NUMTOL, &o));
        if (any_missing) { af_t.assign((size_t)bs * P, 0.0); ns_t.assign((size_t)bs * P, 0); }
        for (int j = 0; j < bs; ++j) {
          ns1[j] = nobs_v[j];
          total[j] = std::nearbyint(mean_v[j] * (double)nobs_v[j]);       // the allele count is an integer: mean = total / n_obs
          if (std::min(total[j], 2.0 * ns1[j] - total[j]) < p.min_mac) variant_ignored[j] = 1;   // compute_mac (Geno.cpp:3077-3108), autosomes
          if (any_missing)                                                  // update_trait_counts (Geno.cpp:2948-2959) as differences from the totals
            for (int q = 0; q < P; ++q) {
              af_t[(size_t)j * P + q] = std::nearbyint(totp_v[(size_t)j * P + q]) - total[j];
              ns_t[(size_t)j * P + q] = (int64_t)nobsp_v[(size_t)j * P + q] - ns1[j];
            }
        }
      } else {
        // parseSnpfromBed: decode the analysed samples, allele counts
        G.assign((size_t)bs * n, 0.0);
        if (any_missing) { af_t.assign((size_t)bs * P, 0.0); mac_t.assign((size_t)bs * P, 0.0); ns_t.assign((size_t)bs * P, 0); }
        parallel_for(bs, nthreads, [&](int j) {
          const uint8_t* row = rows.data() + (size_t)j * r.bpr;
          double* g = G.data() + (size_t)j * n;
          double tot = 0.0; int64_t ns = 0;
          for (int64_t k = 0; k < n; ++k) {
            const int64_t i = file_idx[k];
            double hc = lut[(row[i >> 2] >> (2 * (i & 3))) & 3];
            if (flip && hc != -3.0) hc = 2.0 - hc;
            g[k] = hc;
            if (hc != -3.0) {
              tot += hc; ++ns;
              if (any_missing && has_missing[k])   // update_trait_counts (Geno.cpp:2948-2959): subtract from the totals of the traits the sample is masked for
                for (int q = 0; q < P; ++q)
                  if (!Mc[(size_t)q * n + k]) { af_t[(size_t)j * P + q] -= hc; mac_t[(size_t)j * P + q] -= hc; ns_t[(size_t)j * P + q] -= 1; }
            }
          }
          total[j] = tot; ns1[j] = ns;
          // compute_mac (Geno.cpp:3077-3108), autosomes
          const double mac = std::min(tot, 2.0 * ns - tot);
          if (mac < p.min_mac) variant_ignored[j] = 1;
        });
        s2check(rg_s2_qt_block(s2, G.data(), n, bs, 0, NUMTOL, &o));
      }
      // the result lines (compute_score_qt after the statistic, Step2_Models.cpp:440-466; print_sum_stats_single): formatted by the host threads
      // in contiguous chunks of variants, appended to the files in order
      auto t_fmt = std::chrono::steady_clock::now();
      ms_device += std::chrono::duration<double, std::milli>(t_fmt - t_dev).count();
      const int nchunk = std::max(1, std::min(nthreads, bs / 64));
      std::vector<std::string> chunk_out((size_t)nchunk * P);
      std::vector<int64_t> c_snps(nchunk, 0), c_tests(nchunk, 0), c_tested(nchunk, 0);
      parallel_for(nchunk, nchunk, [&](int t) {
        for (int j = (int)((int64_t)bs * t / nchunk), je = (int)((int64_t)bs * (t + 1) / nchunk); j < je; ++j) {
          if (!variant_ignored[j] && show_info && p.set_min_info && ns1[j] > 0) {   // the all-sample info score below --minINFO drops the variant (Geno.cpp:2349-2353)
            const double af1 = total[j] / (2.0 * ns1[j]);
            double info1 = 1.0;
            if (af1 != 0.0 && af1 != 1.0)
              info1 = r.bgenh ? 1.0 - info_num[j] / (2.0 * ns1[j] * af1 * (1.0 - af1)) : (info_num[j] / ns1[j] - 4.0 * af1 * af1) / (2.0 * af1 * (1.0 - af1));
            if (info1 < p.min_info) variant_ignored[j] = 1;
          }
          if (variant_ignored[j] || ign[j]) { ++c_snps[t]; continue; }
          const int64_t sj = snps[j0 + j];
          std::ostringstream head;
          head << r.snp_chrom[sj] << " " << r.snp_pos[sj] << " " << r.snp_ids[sj] << " " << r.snp_a0[sj] << " " << r.snp_a1[sj] << " ";
          for (int q = 0; q < P; ++q) {
            double af = total[j] / (2.0 * ns1[j]);
            int64_t nsq = ns1[j];
            double infq = show_info ? info_num[j] : 0.0;
            if (test_ignored[(size_t)j * P + q]) continue;
            if (any_missing || glm) {   // compute_mac / compute_aaf_info per trait
              const double tq = total[j] + af_t[(size_t)j * P + q];
              nsq = ns1[j] + ns_t[(size_t)j * P + q];
              const double macq = std::min(tq, 2.0 * nsq - tq);
              if (macq < p.min_mac) { ++c_tests[t]; continue; }
              af = tq / (2.0 * nsq);
              if (show_info) infq += info_t[(size_t)j * P + q];
            }
            double info = 1.0;     // compute_aaf_info (Geno.cpp:3132-3141): IMPUTE info for .bgen, MaCH r2 for .pgen dosages
            if (show_info && af != 0.0 && af != 1.0)
              info = r.bgenh ? 1.0 - infq / (2.0 * nsq * af * (1.0 - af)) : (infq / nsq - 4.0 * af * af) / (2.0 * af * (1.0 - af));
            if (show_info && p.set_min_info && info < p.min_info) { ++c_tests[t]; continue; }     // ignored_trait (Geno.cpp:3143-3144)
            const double st = stats[(size_t)j * P + q];
            double bh = bhat[(size_t)j * P + q], se = bh / st, chisq = st * st;
            bool test_fail = false;
            double logp_spa = -1.0;
            if (correct && corrected[(size_t)j * P + q]) {
              if (corr_fail[(size_t)j * P + q]) test_fail = true;                    // get_sumstats(true, ...): the score test's BETA / SE, no p-value
              else { bh = corr_beta[(size_t)j * P + q]; se = corr_se[(size_t)j * P + q]; chisq = corr_chisq[(size_t)j * P + q]; logp_spa = corr_logp[(size_t)j * P + q]; }
            }
            const double logp = logp_spa >= 0 ? logp_spa : get_logp(chisq);       // --spa prints the p-value it computed, the chi-square is derived from it
            std::ostringstream ln;
            if (af >= 0) ln << head.str() << af << " ";            // print_sum_stats_single (Step2_Models.cpp:2505-2518): a negative value is "NA"
            else ln << head.str() << "NA ";
            if (show_info) { if (info >= 0) ln << info << " "; else ln << "NA "; }      // (the IMPUTE score of very uncertain dosages can be negative)
            ln << nsq << " ADD ";
            if (se >= 0 && !std::isnan(se)) ln << bh << ' ' << se;
            else ln << "NA NA";
            if (chisq >= 0 && !std::isnan(logp) && !test_fail) ln << ' ' << chisq << ' ' << logp;
            else ln << " NA NA";
            ln << (test_fail ? " TEST_FAIL\n" : " NA\n");
            chunk_out[(size_t)t * P + q] += ln.str();
            ++c_tested[t];
          }
        }
      });
      for (int t = 0; t < nchunk; ++t) {
        for (int q = 0; q < P; ++q) *ofs[q] << chunk_out[(size_t)t * P + q];
        n_ignored_snps += c_snps[t]; n_ignored_tests += c_tests[t]; n_tested += c_tested[t];
      }
      ms_format += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fmt).count();
      sout << "done (" << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t1).count() << "ms) \n";
    }
  }
  cleanup.run();
  if (bdev) {
    if (getenv("RG_TIMING"))
      fprintf(stderr, "[timing] step 2 part %d: BGEN on the device: %lld blocks (%lld on the host route) | reading the stored streams %.0f ms | copy + inflate + walk on the GPU %.0f ms (both overlapped with the tests of the previous block)\n",
              part.part, (long long)n_dev_blocks, (long long)n_host_blocks, ms_dev_read, ms_dev_decode);
    rg_bgen_dev_destroy(bdev);
    bdev = nullptr;
  }
  if (getenv("RG_TIMING"))
    fprintf(stderr, "[timing] step 2 part %d: host threads %d (read-ahead %d) | chromosome set-up %.0f ms | waiting for the prepared block %.0f ms (preparing: %.0f ms wall, overlapped; %.0f thread-ms inflate + %.0f thread-ms byte walk) | "
            "upload + device + results %.0f ms | formatting + writing %.0f ms\n", part.part, nthreads, nt_prep, ms_chr, ms_prep_wait, ms_prep_wall, ms_inflate, ms_walk, ms_device, ms_format);
  if (fd >= 0) close(fd);
  rg_s2_destroy(s2);
  part.n_ignored_snps = n_ignored_snps; part.n_ignored_tests = n_ignored_tests;
  part.firth_body = firth_file_body;
  return 0;
}

// `--step 2` on G GPUs (G = 1: the calling thread): contiguous block ranges per GPU, floor(B / G) blocks each and the first B mod G one more
// (the split of write_l0_master, Data.cpp:270-302), one host thread and one library context per GPU, no collective on the data path.
int run_step2_all(Run& r, std::chrono::steady_clock::time_point t_start) {
  const Params& p = r.p;
  const int P = r.P, G = p.gpus;
  std::map<int, int64_t> cn;
  for (int c : r.snp_chrom) cn[c]++;
  int B = 0;
  for (auto& kv : cn) B += (int)((kv.second + p.bsize - 1) / p.bsize);
  std::vector<S2Part> parts(G);
  int b0 = 0;
  for (int g = 0; g < G; ++g) {
    parts[g].part = g; parts[g].nparts = G; parts[g].device = p.single_device ? p.device : p.device + g;
    parts[g].blk_lo = b0; b0 += B / G + (g < B % G ? 1 : 0); parts[g].blk_hi = b0;
  }
  if (G == 1) { parts[0].blk_hi = INT_MAX; run_step2(r, t_start, parts[0]); }
  else {
    sout << std::left << std::setw(20) << " * # GPUs" << ": [" << G << "] (blocks [1.." << B << "] in contiguous ranges)\n";
    std::vector<std::ostringstream> logs(G);
    std::vector<std::exception_ptr> errs(G, nullptr);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g)
      th.emplace_back([&, g]() {
        tl_log = &logs[g];
        try { run_step2(r, t_start, parts[g]); } catch (...) { errs[g] = std::current_exception(); }
        tl_log = nullptr;
      });
    for (auto& t : th) t.join();
    for (int g = 0; g < G; ++g) {
      sout << " GPU " << g << " : blocks [" << parts[g].blk_lo + 1 << ".." << parts[g].blk_hi << "]\n";
      if (g == 0) sout << logs[g].str();
      else {   // the run-wide header lines were logged by part 0
        const std::string lg = logs[g].str();
        const size_t at = lg.find("Chromosome ");
        if (at != std::string::npos) sout << lg.substr(at);
      }
    }
    for (int g = 0; g < G; ++g)
      if (errs[g]) {   // a part failed: no partial result files are left behind
        for (int h = 0; h < G; ++h) for (auto& f : parts[h].files) if (!f.empty()) std::remove(f.c_str());
        std::rethrow_exception(errs[g]);
      }
    // the parts' result files in block order -> PFX_<trait>.regenie[.gz]
    for (int q = 0; q < P; ++q) {
      const std::string fn = p.out + "_" + r.pheno_names[q] + ".regenie" + (p.gz ? ".gz" : "");
      TextOut of(fn, p.gz);
      if (!of) throw std::runtime_error("cannot write file : " + fn);
      std::vector<char> buf(8 << 20);
      for (int g = 0; g < G; ++g) {
        std::ifstream in(parts[g].files[q], std::ios::binary);
        if (!in) throw std::runtime_error("cannot read the results of GPU " + std::to_string(g) + " : " + parts[g].files[q]);
        while (in) { in.read(buf.data(), (std::streamsize)buf.size()); of.write(buf.data(), in.gcount()); }
        if (in.bad() || !of) throw std::runtime_error("error while merging " + parts[g].files[q] + " into " + fn + " (disk full?)");
        in.close();
      }
      of.flush();
      if (!of) throw std::runtime_error("error while writing file : " + fn + " (disk full?)");
      for (int g = 0; g < G; ++g) std::remove(parts[g].files[q].c_str());    // only once the merged file is complete
      parts[0].files[q] = fn;
    }
  }
  if (p.write_null_firth) {   // print_null_firth_info (Step2_Models.cpp:1871-1900): PFX_<k>.firth per trait + PFX_firth.list
    std::ofstream fl(p.out + "_firth.list");
    for (int q = 0; q < P; ++q) {
      std::string body;
      std::set<std::string> seen;     // a chromosome that spans two parts was fitted by both: one line
      for (int g = 0; g < G; ++g) {
        std::istringstream is(parts[g].firth_body.empty() ? std::string() : parts[g].firth_body[q]);
        std::string ln;
        while (std::getline(is, ln)) { const std::string chr = ln.substr(0, ln.find(' ')); if (seen.insert(chr).second) body += ln + "\n"; }
      }
      if (body.empty()) continue;
      const std::string ffn = p.out + "_" + std::to_string(q + 1) + ".firth" + (p.gz ? ".gz" : "");
      TextOut ff(ffn, p.gz);
      if (!ff) throw std::runtime_error("cannot write file : " + ffn);
      ff << body;
      fl << r.pheno_names[q] << " " << (p.use_rel_path ? ffn : get_fullpath(ffn)) << "\n";
    }
    sout << "List of files with null Firth estimates written to: [" << p.out << "_firth.list]\n";
  }
  int64_t n_ignored = 0;
  for (auto& pt : parts) n_ignored += pt.n_ignored_snps * P + pt.n_ignored_tests;
  sout << "\nAssociation results stored separately for each trait in files : \n";
  for (auto& fn : parts[0].files) sout << "* [" << fn << "]\n";
  sout << "\nNumber of ignored tests due to low MAC" << (p.set_min_info ? " or info score" : "") << " : " << n_ignored << "\n";
  sout << "\nElapsed time : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << "s\nEnd of run\n";
  return 0;
}

}  // namespace rgdrv
