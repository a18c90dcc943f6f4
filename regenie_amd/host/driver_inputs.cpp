// regenie-amd, the C++ host driver (see driver.h): genotype metadata, phenotype / covariate files, LOCO files, level-0 job files.
#include "driver.h"

namespace rgdrv {

// prep_bgen (Geno.cpp:38-175): variant list from the file itself, sample identifiers embedded or from --sample
void read_bgen_meta(Run& r) {
  const Params& p = r.p;
  sout << std::left << std::setw(20) << " * bgen" << ": [" << p.bgen << "]\n";
  if (rg_bgen_open(&r.bgenh, p.bgen.c_str()) != RG_BGEN_OK) {
    const std::string msg = rg_bgen_last_error(r.bgenh);
    rg_bgen_close(r.bgenh);
    r.bgenh = nullptr;
    throw std::runtime_error(msg);
  }
  int64_t ns = 0, nv = 0;
  int32_t comp = 0, has_ids = 0;
  rg_bgen_info(r.bgenh, &ns, &nv, &comp, &has_ids);
  sout << "   -summary : bgen file (v1.2 layout, " << (comp == 1 ? "zlib " : comp == 2 ? "zstd " : "un") << "compressed) with " << ns << " "
       << (has_ids ? "named" : "anonymous") << " samples and " << nv << " variants with 8-bit encoding.\n";
  {
    int nt = p.threads;
    if (nt < 1) nt = std::max(1, usable_cpus() - 1);
    rg_bgen_set_threads(r.bgenh, std::min(nt, 64));
  }
  std::set<std::string> ext, exc;
  std::vector<std::string> extract_files = p.extract, exclude_files = p.exclude;
  if (p.run_l0) { extract_files.assign(1, r.job_prefix + ".snplist"); exclude_files.clear(); }
  if (!extract_files.empty()) ext = read_snp_files(extract_files);
  if (!exclude_files.empty()) exc = read_snp_files(exclude_files);
  for (int64_t j = 0; j < nv; ++j) {
    const char *chrom, *rsid, *al0, *al1;
    uint32_t position = 0;
    rg_bgen_variant(r.bgenh, j, &chrom, &position, &rsid, &al0, &al1, nullptr);
    const int c = chr_str_to_int(chrom, p.nchrom);
    if (c == -1) throw std::runtime_error("unknown chromosome code in bgen file.");
    if (r.chr_read.empty() || c != r.chr_read.back()) r.chr_read.push_back(c);
    bool keep = true;
    if (!extract_files.empty() && !ext.count(rsid)) keep = false;
    if (!exclude_files.empty() && exc.count(rsid)) keep = false;
    if (keep) {
      r.snp_chrom.push_back(c); r.snp_offset.push_back(j); r.snp_ids.push_back(rsid);
      if (p.step == 2) {   // prep_bgen (Geno.cpp:80-86): allele0 is the file's second allele unless --ref-first ("switch so allele0 is ALT")
        r.snp_pos.push_back((int64_t)position);
        r.snp_a0.push_back(p.ref_first ? al0 : al1);
        r.snp_a1.push_back(p.ref_first ? al1 : al0);
      }
    }
  }
  sout << "   -n_snps = " << nv << "\n";
  if (!extract_files.empty()) sout << "   -keeping variants specified by --extract\n";
  if (!exclude_files.empty()) sout << "   -removing variants specified by --exclude\n";
  if (r.snp_chrom.empty()) throw std::runtime_error("no variant left to include in analysis.");
  if (r.snp_chrom.size() > 1000000 && !p.force_step1)
    throw std::runtime_error("it is not recommened to use more than 1M variants in step 1 (use --force-step1 to override)");
  // samples
  if (!p.sample_file.empty()) {  // read_bgen_sample (Geno.cpp:395-456)
    std::string fn = p.sample_file;
    if (!file_exists(fn)) fn += ".gz";
    sout << "   -sample file: " << fn << "\n";
    TextIn f(fn);
    if (!f) throw std::runtime_error("cannot open file : " + fn);
    std::string line;
    int nline = 0;
    while (std::getline(f, line)) {
      auto t = split_ws(line);
      if (t.size() < 2) throw std::runtime_error("incorrectly formatted sample file at line" + std::to_string(r.fam_ids.size() + 1));
      if (nline == 0) { if (t[0] != "ID_1" || t[1] != "ID_2") throw std::runtime_error("header of the sample file must start with: ID_1 ID_2"); }
      else if (nline == 1) { if (t[0] != "0" || t[1] != "0") throw std::runtime_error("second line of sample file must start with: 0 0."); }
      else {
        r.fam_ids.push_back(t[0] + "_" + t[1]);
        if (t.size() >= 4 && t[3] != "0" && t[3] != "NA" && t[3] != "1" && t[3] != "2") throw std::runtime_error("unrecognized sex code in file : '" + t[3] + "'");
      }
      ++nline;
    }
    if ((int64_t)r.fam_ids.size() != ns) throw std::runtime_error("number of samples in BGEN file does not match that in the sample file.");
  } else {
    if (!has_ids) throw std::runtime_error("bgen file has no sample identifiers; specify a sample file with --sample");
    for (int64_t i = 0; i < ns; ++i) {
      const char* id;
      rg_bgen_sample_id(r.bgenh, i, &id);
      r.fam_ids.push_back(id);
    }
  }
  {
    std::set<std::string> seen;
    for (auto& id : r.fam_ids)
      if (!seen.insert(id).second) throw std::runtime_error("duplicate individual in bgen file : FID_IID =" + id);
  }
  r.n_file = ns;
  sout << "   -n_samples = " << ns << "\n";
  r.bpr = (r.n_file + 3) / 4;
  r.dosage_mode = true;
  apply_sample_and_variant_filters(r);
}

void read_bim_fam(Run& r) {  // bed: Geno.cpp:518-610, :643-690, :1128-1220; pgen: read_pvar / read_psam, Geno.cpp:771-1004
  const Params& p = r.p;
  if (!p.bgen.empty()) { read_bgen_meta(r); return; }
  const bool pg = !p.pgen.empty();
  std::future<void> fam_task;     // the .fam file is read on its own thread while this one reads the .bim (two 500,000-line files at BASELINE configs[2])
  std::string fam_fn;
  if (!pg) {
    fam_fn = p.bed + ".fam";
    {
      std::ifstream f(fam_fn);
      if (!f) throw std::runtime_error("cannot open file : " + fam_fn);
    }
    fam_task = std::async(std::launch::async, [&r, fam_fn]() {
      std::ifstream f(fam_fn);
      TextLines lines;
      slurp_lines(f, lines);
      std::unordered_set<std::string> seen;
      seen.reserve(lines.size() * 2);
      r.fam_ids.reserve(lines.size());
      Tok t[6];
      for (size_t li = 0; li < lines.size(); ++li) {
        if (tokenize(lines.begin(li), lines.end(li), t, 6) < 6) throw std::runtime_error("incorrectly formatted fam file at line " + std::to_string(r.fam_ids.size() + 1));
        std::string id;
        id.reserve((size_t)(t[0].e - t[0].b) + 1 + (size_t)(t[1].e - t[1].b));
        id.append(t[0].b, t[0].e).push_back('_');
        id.append(t[1].b, t[1].e);
        if (!seen.insert(id).second) throw std::runtime_error("duplicate individual in fam file : FID_IID=" + id);
        const bool one = t[4].e - t[4].b == 1;
        if (!one || (*t[4].b != '0' && *t[4].b != '1' && *t[4].b != '2')) throw std::runtime_error("unrecognized sex code in file : '" + std::string(t[4].b, t[4].e) + "'");
        if (*t[4].b == '1') r.has_male = true;
        r.fam_ids.push_back(std::move(id));
      }
      r.n_file = (int64_t)r.fam_ids.size();
    });
  } else {  // read_psam (Geno.cpp:941-1004): header line "#FID IID [SEX ...]", any "##" lines before it are skipped
    std::string fn = p.pgen + ".psam";
    if (!file_exists(fn)) fn += ".gz";  // Geno.cpp:952
    TextIn f(fn);
    if (!f) throw std::runtime_error("cannot open file : " + fn);
    sout << std::left << std::setw(20) << " * psam" << ": [" << fn << "] ";
    std::string line;
    std::vector<std::string> t;
    while (std::getline(f, line)) {
      t = split_ws(line);
      if (t.empty()) throw std::runtime_error("no blank lines should be before the header line in psam file.");
      if (t[0] == "#IID") throw std::runtime_error("invalid header (must start with #FID [not #IID]).");
      if (t[0] == "#FID") break;
    }
    if (t.size() < 2 || t[1] != "IID") throw std::runtime_error("header does not have the correct format.");
    const auto sc = std::find(t.begin(), t.end(), "SEX");
    const bool has_sex = sc != t.end();
    const size_t sex_col = has_sex ? (size_t)(sc - t.begin()) : 0;
    std::set<std::string> seen;
    while (std::getline(f, line)) {
      t = split_ws(line);
      if (t.size() < 3) throw std::runtime_error("incorrectly formatted psam file at line " + std::to_string(r.fam_ids.size() + 1));
      std::string id = t[0] + "_" + t[1];
      if (!seen.insert(id).second) throw std::runtime_error("duplicate individual in fam file : FID_IID=" + id);
      if (has_sex) {
        if (sex_col >= t.size()) throw std::runtime_error("incorrectly formatted psam file at line " + std::to_string(r.fam_ids.size() + 1));
        const std::string& sx = t[sex_col];
        if (sx != "0" && sx != "NA" && sx != "1" && sx != "2") throw std::runtime_error("unrecognized sex code in file : '" + sx + "'");
        if (sx == "1") r.has_male = true;
      }
      r.fam_ids.push_back(id);
    }
    r.n_file = (int64_t)r.fam_ids.size();
    sout << "n_samples = " << r.n_file << "\n";
  }
  std::set<std::string> ext, exc;
  std::vector<std::string> extract_files = p.extract, exclude_files = p.exclude;
  if (p.run_l0) {  // the job's snplist replaces --extract; --exclude is ignored (Data.cpp:852-854, Regenie.cpp:668)
    extract_files.assign(1, r.job_prefix + ".snplist");
    exclude_files.clear();
  }
  if (!extract_files.empty()) ext = read_snp_files(extract_files);
  if (!exclude_files.empty()) exc = read_snp_files(exclude_files);
  int64_t n_variants_file = 0;
  {
    const std::string kind = pg ? "pvar" : "bim";
    std::string fn = pg ? p.pgen + ".pvar" : p.bed + ".bim";
    if (pg && !file_exists(fn)) fn += ".gz";  // Geno.cpp:783
    TextIn f(fn);
    if (!f) throw std::runtime_error("cannot open file : " + fn);
    std::ostringstream head;                       // printed once the .fam thread has delivered its own line
    head << std::left << std::setw(20) << (" * " + kind) << ": [" << fn << "] ";
    std::string line;
    int64_t lineno = 0;
    int minchr = 0;
    size_t min_cols = 6, id_col = 1, pos_col = 3, ref_col = 0, alt_col = 0;
    if (pg) {  // read_pvar (Geno.cpp:787-815): skip to the "#CHROM" header and locate the POS / ID / REF / ALT columns
      std::vector<std::string> t;
      while (std::getline(f, line)) {
        t = split_ws(line);
        if (t.empty()) throw std::runtime_error("no blank lines should be before the header line in pvar file.");
        if (t[0] == "#CHROM") break;
      }
      if (t.size() < 5) throw std::runtime_error("header of pvar file does not have correct format.");
      const auto idc = std::find(t.begin(), t.end(), "ID");
      for (const char* col : {"POS", "ID", "REF", "ALT"})
        if (std::find(t.begin(), t.end(), col) == t.end()) throw std::runtime_error("header of pvar file does not have correct format.");
      min_cols = 5;
      id_col = (size_t)(idc - t.begin());
      pos_col = (size_t)(std::find(t.begin(), t.end(), "POS") - t.begin());
      ref_col = (size_t)(std::find(t.begin(), t.end(), "REF") - t.begin());
      alt_col = (size_t)(std::find(t.begin(), t.end(), "ALT") - t.begin());
    }
    while (std::getline(f, line)) {
      auto t = split_ws(line);
      if (t.size() < min_cols || id_col >= t.size())
        throw std::runtime_error("incorrectly formatted " + kind + " file at line " + std::to_string(lineno + 1));
      int c = chr_str_to_int(t[0], p.nchrom);
      if (c == -1) throw std::runtime_error("unknown chromosome code in " + kind + " file at line " + std::to_string(lineno + 1));
      if (r.chr_read.empty() || c != r.chr_read.back()) {
        r.chr_read.push_back(c);
        if (c <= minchr) throw std::runtime_error("chromosomes in " + kind + " file are not in ascending order.");
        minchr = c;
      }
      const std::string& vid = t[id_col];
      bool keep = true;
      if (!extract_files.empty() && !ext.count(vid)) keep = false;
      if (!exclude_files.empty() && exc.count(vid)) keep = false;
      if (keep) {
        r.snp_chrom.push_back(c); r.snp_offset.push_back(lineno); r.snp_ids.push_back(vid);
        if (!pg && p.step == 2) {   // read_bim (Geno.cpp:546-553): the reference allele is the LAST one unless --ref-first
          r.snp_pos.push_back((int64_t)std::strtoul(t[3].c_str(), nullptr, 0));
          r.snp_a0.push_back(p.ref_first ? t[4] : t[5]);
          r.snp_a1.push_back(p.ref_first ? t[5] : t[4]);
        } else if (pg && p.step == 2) {   // read_pvar (Geno.cpp:824-828): allele1 = REF, allele2 = ALT, whatever --ref-first says
          if (std::max(pos_col, std::max(ref_col, alt_col)) >= t.size())
            throw std::runtime_error("incorrectly formatted " + kind + " file at line " + std::to_string(lineno + 1));
          r.snp_pos.push_back((int64_t)std::strtoul(t[pos_col].c_str(), nullptr, 0));
          r.snp_a0.push_back(t[ref_col]);
          r.snp_a1.push_back(t[alt_col]);
        }
      }
      ++lineno;
    }
    n_variants_file = lineno;
    if (fam_task.valid()) {
      fam_task.get();                              // rethrows the .fam reader's error
      sout << std::left << std::setw(20) << " * fam" << ": [" << fam_fn << "] " << "n_samples = " << r.n_file << "\n";
    }
    sout << head.str() << "n_snps = " << lineno << "\n";
    if (!extract_files.empty()) sout << "   -keeping variants specified by --extract\n";
    if (!exclude_files.empty()) sout << "   -removing variants specified by --exclude\n";
    if (r.snp_chrom.empty()) throw std::runtime_error("no variant left to include in analysis.");
    if (!extract_files.empty() || !exclude_files.empty())
      sout << "   -number of variants remaining in the analysis = " << r.snp_chrom.size() << "\n";
  }
  if (p.step == 1 && r.snp_chrom.size() > 1000000 && !p.force_step1)  // Data.cpp:173-175
    throw std::runtime_error("it is not recommened to use more than 1M variants in step 1 (use --force-step1 to override)");
  if (!pg) {
    std::string fn = p.bed + ".bed";
    std::ifstream f(fn, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open file : " + fn);
    unsigned char magic[3];
    f.read((char*)magic, 3);
    if (magic[0] != 0x6c || magic[1] != 0x1b || magic[2] != 0x01) throw std::runtime_error("invalid bed file format.");
    sout << std::left << std::setw(20) << " * bed" << ": [" << fn << "]\n";
    r.bpr = (r.n_file + 3) / 4;
  } else {  // prep_pgen (Geno.cpp:1071-1103)
    std::string fn = p.pgen + ".pgen";
    sout << std::left << std::setw(20) << " * pgen" << ": [" << fn << "] \n";
    if (rg_pgen_open(&r.pgen, fn.c_str()) != RG_PGEN_OK) {
      const std::string msg = rg_pgen_last_error(r.pgen);
      rg_pgen_close(r.pgen);
      r.pgen = nullptr;
      throw std::runtime_error(msg);
    }
    {  // --threads, default = hardware threads - 1 (Regenie.cpp:1104-1106); the decode of a block's variants is spread over them
      int nt = p.threads;
      if (nt < 1) nt = std::max(1, usable_cpus() - 1);
      rg_pgen_set_threads(r.pgen, std::min(nt, 64));
    }
    int64_t ns = 0, nv = 0;
    int32_t has_dosage = 0;
    rg_pgen_info(r.pgen, &ns, &nv, nullptr, nullptr, &has_dosage);
    r.dosage_mode = has_dosage != 0;  // params->dosage_mode (Geno.cpp:1101): every variant is then read with Read(), not ReadHardcalls()
    if (r.dosage_mode) sout << "   -dosages present: level 0 runs on the fp64 genotype path\n";
    if (ns != r.n_file) throw std::runtime_error("number of samples in pgen file and psam file don't match.");
    if (nv != n_variants_file) throw std::runtime_error("number of variants in pgen file and pvar file don't match.");
    r.bpr = (r.n_file + 3) / 4;
  }
  apply_sample_and_variant_filters(r);
}

void apply_sample_and_variant_filters(Run& r) {
  const Params& p = r.p;
  // --keep / --remove (Geno.cpp:1263-1341)
  r.ind_ignore.assign(r.n_file, 0);
  if (!p.remove.empty()) {
    auto s = read_id_files(p.remove);
    sout << "   -removing individuals specified by --remove\n";
    for (int64_t i = 0; i < r.n_file; ++i) r.ind_ignore[i] = s.count(r.fam_ids[i]) ? 1 : 0;
  } else if (!p.keep.empty()) {
    auto s = read_id_files(p.keep);
    sout << "   -keeping only individuals specified by --keep\n";
    for (int64_t i = 0; i < r.n_file; ++i) r.ind_ignore[i] = s.count(r.fam_ids[i]) ? 0 : 1;
  }
  for (int64_t i = 0; i < r.n_file; ++i)
    if (!r.ind_ignore[i]) r.ids.push_back(r.fam_ids[i]);
  r.N = (int64_t)r.ids.size();
  if (r.N == 0) throw std::runtime_error("no samples remaining in the analysis.");
  if (r.N != r.n_file) sout << "   -number of genotyped individuals remaining in the analysis = " << r.N << "\n";
}

// --pred list + first pass over every LOCO file (check_blup / blup_read, Pheno.cpp:1204-1391): header ids -> samples,
// line 2 tells which samples have NA predictions (masked for the trait), byte offsets of the chromosome lines for later
void blup_read(Run& r, const std::unordered_map<std::string, int64_t>& idx) {
  const Params& p = r.p;
  const int64_t N = r.N;
  std::map<std::string, std::string> files;
  {
    TextIn f(p.pred_list);
    if (!f) throw std::runtime_error("cannot open file : " + p.pred_list);
    std::string line;
    while (std::getline(f, line)) {
      auto t = split_ws(line);
      if (t.empty()) continue;
      if (t.size() != 2) throw std::runtime_error("step 1 list file is not in the right format : " + p.pred_list);
      if (files.count(t[0])) throw std::runtime_error("phenotype '" + t[0] + "' appears more than once in step 1 list file.");
      files[t[0]] = t[1];
    }
  }
  sout << " * LOCO predictions : [" << p.pred_list << "]\n";
  r.blups.resize(r.P);
  // one host thread per phenotype (each file is ~115 MB at 500,000 samples x 23 rows); messages and errors in phenotype order
  std::vector<std::string> logs(r.P), errs(r.P);
  const int nt_files = std::max(1, std::min<int>(r.P, std::min(32, usable_cpus())));
  parallel_for(r.P, nt_files, [&](int q) {
    try {
    std::ostringstream lg;
    if (!files.count(r.pheno_names[q])) throw std::runtime_error("No step 1 file provided for phenotype '" + r.pheno_names[q] + "'.");
    Run::Blup& bl = r.blups[q];
    bl.file = files.at(r.pheno_names[q]);
    lg << "   -file [" << bl.file << "] for phenotype '" << r.pheno_names[q] << "'\n";
    // a gzipped file (`--step 1 --gz` writes PFX_<k>.loco.gz and lists it; Files::openForRead inflates it) cannot be revisited by byte
    // offset: its chromosome rows (nChrom lines) are kept in memory instead
    const bool gzf = ends_with_gz(bl.file);
    TextIn fgz(gzf ? bl.file : std::string("/dev/null"));
    std::ifstream fpl;
    if (!gzf) fpl.open(bl.file, std::ios::binary);
    std::istream& f = gzf ? static_cast<std::istream&>(fgz) : static_cast<std::istream&>(fpl);
    if (!f) throw std::runtime_error("cannot open file : " + bl.file);
    std::string line;
    std::getline(f, line);
    std::vector<Tok> hdr(line.size() / 2 + 2);
    const int nh = tokenize(line.data(), line.data() + line.size(), hdr.data(), (int)hdr.size());
    if (nh == 0 || std::string(hdr[0].b, hdr[0].e) != "FID_IID") throw std::runtime_error("header of blup file must start with FID_IID (=" + (nh == 0 ? std::string() : std::string(hdr[0].b, hdr[0].e)) + ")");
    bl.col_sample.assign(nh, -1);
    {
      std::string key;
      for (int c = 1; c < nh; ++c) {
        key.assign(hdr[c].b, hdr[c].e);
        auto it = idx.find(key);
        if (it != idx.end()) bl.col_sample[c] = it->second;
      }
    }
    bl.line_off.push_back(gzf ? 0 : (int64_t)f.tellg());
    std::string line2;
    std::getline(f, line2);
    if (gzf) bl.lines.push_back(line2);
    std::vector<Tok> l2(nh + 1);
    if (tokenize(line2.data(), line2.data() + line2.size(), l2.data(), (int)l2.size()) != nh)
      throw std::runtime_error("blup file for phenotype '" + r.pheno_names[q] + "' has different number of entries on line 2 compared to the header.");
    std::vector<uint8_t> have(N, 0);
    for (int c = 1; c < nh; ++c)
      if (bl.col_sample[c] >= 0 && convert_double_tok(l2[c].b, l2[c].e) != MISSING) have[bl.col_sample[c]] = 1;
    int64_t before = 0, after = 0;
    for (int64_t i = 0; i < N; ++i) { before += r.mask[(size_t)q * N + i]; r.mask[(size_t)q * N + i] &= have[i]; after += r.mask[(size_t)q * N + i]; }
    if (after < 1) throw std::runtime_error("all individuals are missing LOCO predictions for phenotype '" + r.pheno_names[q] + "'.");
    if (after < before) lg << "    + " << before - after << " individuals with missing LOCO predictions will be ignored for the trait\n";
    for (;;) {   // offsets of the following lines (one per chromosome)
      const int64_t off = gzf ? 0 : (int64_t)f.tellg();
      if (!std::getline(f, line) || line.empty()) break;
      bl.line_off.push_back(off);
      if (gzf) bl.lines.push_back(line);
    }
    logs[q] = lg.str();
    } catch (const std::exception& e) { errs[q] = e.what(); if (errs[q].empty()) errs[q] = "cannot read blup file"; }
  });
  for (int q = 0; q < r.P; ++q) {
    if (!errs[q].empty()) throw std::runtime_error(errs[q]);
    sout << logs[q];
  }
}

void read_pheno_cov(Run& r) {  // Pheno.cpp:50-146, :148-364, :573-808, :810-841, :1903-1935
  const Params& p = r.p;
  const int64_t N = r.N;
  const bool tmk = getenv("RG_TIMING") != nullptr;
  auto tm0 = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!tmk) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[timing] read_pheno_cov %-28s %6lld ms\n", what, (long long)std::chrono::duration_cast<std::chrono::milliseconds>(t - tm0).count());
    tm0 = t;
  };
  std::unordered_map<std::string, int64_t> idx;      // blup_read (step 2) looks samples up by their full id string
  if (p.step == 2) {
    idx.reserve((size_t)N * 2);
    for (int64_t i = 0; i < N; ++i) idx[r.ids[i]] = i;
  }
  const IdIndex ids(r.ids);                          // the sample files: FID / IID token pairs -> sample, no key string built
  const int nt_parse = std::max(1, std::min(32, usable_cpus() - 1));
  std::vector<uint8_t> in_pheno(N, 0), in_cov(N, p.covar_file.empty() ? 1 : 0);
  {
    TextIn f(p.pheno_file);
    if (!f) throw std::runtime_error("cannot open file : " + p.pheno_file);
    sout << std::left << std::setw(20) << " * phenotypes" << ": [" << p.pheno_file << "] ";
    std::string line;
    std::getline(f, line);
    auto hdr = split_ws(line);
    if (hdr.size() < 2) throw std::runtime_error("header of phenotype file has too few columns.");
    if (hdr[0] != "FID" || hdr[1] != "IID") throw std::runtime_error("header of phenotype file must start with: FID IID.");
    std::set<std::string> want(p.pheno_cols.begin(), p.pheno_cols.end());
    std::vector<int> keep_cols;
    if (p.t2e) {   // the TIME columns in file order, then their event columns (files->t2e_map, Regenie.cpp:578-585; Pheno.cpp:230-283)
      std::vector<std::pair<int, int>> te;          // (header index of the time column, of its event column)
      std::vector<int> all;
      for (size_t k = 0; k < p.pheno_cols.size(); ++k) {
        int tj = -1, ej = -1;
        for (size_t j = 2; j < hdr.size(); ++j) { if (hdr[j] == p.pheno_cols[k]) tj = (int)j; if (hdr[j] == p.event_cols[k]) ej = (int)j; }
        if (tj < 0 || ej < 0) throw std::runtime_error("time-to-event column '" + (tj < 0 ? p.pheno_cols[k] : p.event_cols[k]) + "' is not in the phenotype file.");
        te.emplace_back(tj, ej);
        all.push_back(tj); all.push_back(ej);
      }
      std::sort(te.begin(), te.end());
      std::sort(all.begin(), all.end());
      for (auto& x : te) { keep_cols.push_back(x.first); r.pheno_names.push_back(hdr[x.first]); r.t2e_num.push_back((int)(std::lower_bound(all.begin(), all.end(), x.first) - all.begin()) + 1); }
      for (auto& x : te) keep_cols.push_back(x.second);
      r.P = (int)te.size();
    } else {
      for (size_t j = 2; j < hdr.size(); ++j)
        if (want.empty() || want.count(hdr[j])) { keep_cols.push_back((int)j); r.pheno_names.push_back(hdr[j]); }
      r.P = (int)keep_cols.size();
    }
    const int NV = (int)keep_cols.size();        // values read per line (--t2e: a time and an event per trait)
    if (r.P < 1) throw std::runtime_error("need at least one phenotype.");
    sout << "n_pheno = " << r.P << "\n";
    const bool strict = p.strict || (r.P == 1 && !p.t2e);  // Pheno.cpp:198 (with --t2e the reference counts 2 columns per trait)
    if (strict) sout << "   -dropping observations with missing values at any of the phenotypes\n";
    else sout << "   -keeping and mean-imputing missing observations (done for each trait)\n";
    r.Y.assign((size_t)N * r.P, 0.0);
    r.mask.assign((size_t)N * r.P, 1);
    if (p.bt || p.ct || p.t2e) r.Yraw.assign((size_t)N * r.P, 0.0);
    if (p.t2e) r.Yevent.assign((size_t)N * r.P, 0.0);
    // The lines are tokenised, matched to their sample and converted by several threads (at 500,000 samples x 10 phenotypes one thread
    // needs 2 s); the checks and the bookkeeping below then run over the records in file order, exactly as a line-by-line reader would.
    mark("id index + header");
    TextLines lines;
    slurp_lines(f, lines);
    mark("pheno: file in memory");
    struct Rec { int64_t i; int state; };      // state 0: use, 1: blank line, 2: wrong number of columns, 3: a value that is not a number
    std::vector<Rec> recs(lines.size());
    std::vector<double> vals(lines.size() * (size_t)NV);
    {
      const int ncol = (int)hdr.size();
      const int nchunk = (int)std::min<size_t>(lines.size(), (size_t)nt_parse * 4);
      parallel_for(nchunk, nt_parse, [&](int c) {
        std::vector<Tok> t((size_t)ncol);
        for (size_t li = lines.size() * c / nchunk, le = lines.size() * (c + 1) / nchunk; li < le; ++li) {
          const int nt = tokenize(lines.begin(li), lines.end(li), t.data(), ncol);
          Rec& rc = recs[li];
          rc.i = -1; rc.state = 0;
          if (nt == 0) { rc.state = 1; continue; }
          if (nt != ncol) { rc.state = 2; continue; }
          rc.i = ids.find(t[0].b, t[0].e, t[1].b, t[1].e);
          if (rc.i < 0) continue;
          try { for (int q = 0; q < NV; ++q) vals[li * (size_t)NV + q] = convert_double_tok(t[keep_cols[q]].b, t[keep_cols[q]].e); }
          catch (...) { rc.state = 3; }
        }
      });
    }
    mark("pheno: parallel parse");
    for (size_t li = 0; li < lines.size(); ++li) {
      if (recs[li].state == 1) continue;
      if (recs[li].state == 2) throw std::runtime_error("incorrectly formatted phenotype file.");
      if (recs[li].i < 0) continue;
      const int64_t i = recs[li].i;
      std::vector<std::string> t;                     // the tokens again, for the messages of the rare failing line only
      auto tok = [&]() -> const std::vector<std::string>& { if (t.empty()) t = split_ws(lines.line(li)); return t; };
      if (recs[li].state == 3) for (int q = 0; q < NV; ++q) (void)convert_double(tok()[keep_cols[q]]);     // rethrows the conversion error
      if (in_pheno[i]) throw std::runtime_error("individual appears more than once in phenotype file: FID=" + tok()[0] + " IID=" + tok()[1]);
      in_pheno[i] = 1;
      bool all_miss = true;
      for (int q = 0; q < r.P && p.t2e; ++q) {   // Pheno.cpp:262-283
        const double tv = vals[li * (size_t)NV + q];
        double ev = vals[li * (size_t)NV + r.P + q];
        if (p.cc12 && ev != MISSING) ev -= 1;
        r.Y[(size_t)q * N + i] = r.Yraw[(size_t)q * N + i] = tv;
        r.Yevent[(size_t)q * N + i] = ev;
        if (tv < 0 && tv != MISSING) throw std::runtime_error("a phenotype time value is <0 for individual: FID=" + tok()[0] + " IID=" + tok()[1] + " Y=" + tok()[keep_cols[q]]);
        if (ev != 0 && ev != 1 && ev != MISSING) throw std::runtime_error("a phenotype censor value is invalid for individual: FID=" + tok()[0] + " IID=" + tok()[1] + " Y=" + tok()[keep_cols[r.P + q]]);
        if (tv != MISSING && ev == MISSING) throw std::runtime_error("a phenotype has missing censor with non-missing time for individual: FID=" + tok()[0] + " IID=" + tok()[1]);
        if (tv == MISSING) { r.mask[(size_t)q * N + i] = 0; r.Yevent[(size_t)q * N + i] = MISSING; }
        else all_miss = false;
      }
      for (int q = 0; q < r.P && !p.t2e; ++q) {
        double v = vals[li * (size_t)NV + q];
        if (p.bt) {  // Pheno.cpp:260-283
          if (p.cc12 && v != MISSING) v -= 1;
          r.Yraw[(size_t)q * N + i] = v;
          if (v != 0 && v != 1) {
            if (v != MISSING) throw std::runtime_error("a phenotype value is not 0/1/NA for individual: FID=" + tok()[0] + " IID=" + tok()[1] + " Y=" + tok()[keep_cols[q]]);
            r.mask[(size_t)q * N + i] = 0;
          }
        } else if (p.ct) {  // Pheno.cpp:298, :313-320: counts must be non-negative
          r.Yraw[(size_t)q * N + i] = v;
          if (v < 0) {
            if (v != MISSING) throw std::runtime_error("a phenotype value is <0 for individual: FID=" + tok()[0] + " IID=" + tok()[1] + " Y=" + tok()[keep_cols[q]]);
            r.mask[(size_t)q * N + i] = 0;
          }
        }
        r.Y[(size_t)q * N + i] = v;
        if (v != MISSING) all_miss = false;
        else if (p.step == 2 && !strict && !p.bt && !p.ct) r.mask[(size_t)q * N + i] = 0;   // rm_missing_qt (Pheno.cpp:328, Regenie.cpp:1086)
        else if (strict) {
          for (int q2 = 0; q2 < r.P; ++q2) r.mask[(size_t)q2 * N + i] = 0;
          all_miss = true;
          break;
        }
      }
      if (all_miss) in_pheno[i] = 0;
    }
    for (int q = 0; q < r.P; ++q) {
      int64_t n = 0;
      for (int64_t i = 0; i < N; ++i) { r.mask[(size_t)q * N + i] &= in_pheno[i]; n += r.mask[(size_t)q * N + i]; }
      if (n == 0) throw std::runtime_error("all individuals have missing/invalid values for phenotype '" + r.pheno_names[q] + "'.");
    }
    if (p.bt) {  // rm_phenoCols (Pheno.cpp:528-570): drop traits with too few cases
      std::vector<int> keepq;
      for (int q = 0; q < r.P; ++q) {
        int64_t ncases = 0;
        for (int64_t i = 0; i < N; ++i) ncases += r.Yraw[(size_t)q * N + i] == 1;      // `(phenotypes_raw == 1).colwise().count()`: the mask is not looked at -- a row that
                                                                                          // --strict dropped still counts for the traits read before its first missing value
        if (ncases >= p.min_case_count) keepq.push_back(q);
        else sout << "   -WARNING: phenotype '" << r.pheno_names[q] << "' has fewer than " << p.min_case_count << " cases and is dropped\n";
      }
      if (keepq.empty()) throw std::runtime_error("all phenotypes have less than " + std::to_string(p.min_case_count) + " cases.");
      if ((int)keepq.size() != r.P) {
        std::vector<double> Y2, R2; std::vector<uint8_t> M2; std::vector<std::string> n2;
        for (int q : keepq) {
          Y2.insert(Y2.end(), r.Y.begin() + (size_t)q * N, r.Y.begin() + (size_t)(q + 1) * N);
          R2.insert(R2.end(), r.Yraw.begin() + (size_t)q * N, r.Yraw.begin() + (size_t)(q + 1) * N);
          M2.insert(M2.end(), r.mask.begin() + (size_t)q * N, r.mask.begin() + (size_t)(q + 1) * N);
          n2.push_back(r.pheno_names[q]);
        }
        r.Y.swap(Y2); r.Yraw.swap(R2); r.mask.swap(M2); r.pheno_names.swap(n2);
        r.P = (int)keepq.size();
        if (!strict)
          for (int64_t i = 0; i < N; ++i) {
            bool any = false;
            for (int q = 0; q < r.P; ++q) any |= r.mask[(size_t)q * N + i] != 0;
            in_pheno[i] &= any;
          }
      }
    }
    int64_t np = 0;
    for (int64_t i = 0; i < N; ++i) np += in_pheno[i];
    sout << "   -number of phenotyped individuals " << (strict ? "with no missing data" : "") << " = " << np << "\n";
  }
  if (p.step == 2) blup_read(r, idx);   // prep_run (Pheno.cpp:1063-1068): samples without LOCO predictions are masked for the trait
  int ncols = 1;
  std::vector<double> Xraw;  // col-major N x ncols
  if (!p.covar_file.empty()) {
  mark("pheno: checks in file order");
    TextIn f(p.covar_file);
    if (!f) throw std::runtime_error("cannot open file : " + p.covar_file);
    sout << std::left << std::setw(20) << " * covariates" << ": [" << p.covar_file << "] ";
    std::string line;
    std::getline(f, line);
    auto hdr = split_ws(line);
    if (hdr.size() < 2 || hdr[0] != "FID" || hdr[1] != "IID") throw std::runtime_error("header of covariate file must start with: FID IID.");
    // cov_colKeep_names (Regenie.cpp:591-619, Pheno.cpp:599-632): name -> quantitative?  --catCovarList names are kept too
    std::map<std::string, bool> colmap;
    for (auto& h : p.covar_cols) colmap[h] = true;
    for (auto& h : p.cat_covar) colmap[h] = false;
    std::vector<int> kc;
    std::vector<uint8_t> is_cat;
    std::vector<std::string> cov_names;
    for (size_t j = 2; j < hdr.size(); ++j) {
      bool keep;
      if (p.covar_cols.empty() && !colmap.count(hdr[j])) { colmap[hdr[j]] = true; keep = true; }
      else keep = colmap.count(hdr[j]) != 0;
      if (keep && std::find(r.pheno_names.begin(), r.pheno_names.end(), hdr[j]) != r.pheno_names.end()) {
        keep = false;  // a covariate that is one of the analysed phenotypes is ignored
        colmap.erase(hdr[j]);
      }
      if (keep) { kc.push_back((int)j); is_cat.push_back(colmap[hdr[j]] ? 0 : 1); cov_names.push_back(hdr[j]); }
    }
    if (colmap.size() != kc.size()) throw std::runtime_error("not all covariates specified are found in the covariate file.");
    std::vector<std::map<std::string, int>> levels(kc.size());  // convertNumLevel (Regenie.cpp:1720-1735): order of appearance
    ncols = 1 + (int)kc.size();
    sout << "n_cov = " << kc.size() << "\n";
    Xraw.assign((size_t)N * ncols, 0.0);
    for (int64_t i = 0; i < N; ++i) Xraw[i] = 1.0;
    // as the phenotype file: tokenised, matched and converted by several threads; the checks, the categorical levels (numbered in order of
    // appearance) and the first-missing-value rule then run over the records in file order, exactly as the line-by-line reader did
    TextLines lines;
    slurp_lines(f, lines);
    const int nkc = (int)kc.size(), ncolf = (int)hdr.size();
    struct CRec { int64_t i; int state; int fail; };   // state 0: use, 1: blank, 2: wrong number of columns; fail: first column whose value is not a number (-1: none)
    std::vector<CRec> crec(lines.size());
    std::vector<double> cval(lines.size() * (size_t)std::max(1, nkc));
    {
      const int nchunk = (int)std::min<size_t>(lines.size(), (size_t)nt_parse * 4);
      parallel_for(nchunk, nt_parse, [&](int c) {
        std::vector<Tok> t((size_t)ncolf);
        for (size_t li = lines.size() * c / nchunk, le = lines.size() * (c + 1) / nchunk; li < le; ++li) {
          const int nt = tokenize(lines.begin(li), lines.end(li), t.data(), ncolf);
          CRec& rc = crec[li];
          rc.i = -1; rc.state = 0; rc.fail = -1;
          if (nt == 0) { rc.state = 1; continue; }
          if (nt != ncolf) { rc.state = 2; continue; }
          rc.i = ids.find(t[0].b, t[0].e, t[1].b, t[1].e);
          if (rc.i < 0) continue;
          for (int cc = 0; cc < nkc; ++cc) {
            if (is_cat[cc]) continue;                    // levels are assigned in file order below
            try { cval[li * (size_t)nkc + cc] = convert_double_tok(t[kc[cc]].b, t[kc[cc]].e); }
            catch (...) { rc.fail = cc; break; }
          }
        }
      });
    }
    for (size_t li = 0; li < lines.size(); ++li) {
      if (crec[li].state == 1) continue;
      if (crec[li].state == 2) throw std::runtime_error("incorrectly formatted covariate file.");
      if (crec[li].i < 0) continue;
      const int64_t i = crec[li].i;
      std::vector<std::string> t;
      auto tok = [&]() -> const std::vector<std::string>& { if (t.empty()) t = split_ws(lines.line(li)); return t; };
      if (in_cov[i]) throw std::runtime_error("individual appears more than once in covariate file: FID=" + tok()[0] + " IID=" + tok()[1]);
      in_cov[i] = 1;
      for (int c = 0; c < nkc; ++c) {
        double v;
        if (is_cat[c]) {
          const std::string& tk = tok()[kc[c]];
          if (tk == "NA" || tk == "nan" || tk == "inf") v = MISSING;
          else {
            auto lv = levels[c].find(tk);
            if (lv == levels[c].end()) lv = levels[c].emplace(tk, (int)levels[c].size()).first;
            v = lv->second;
          }
        } else if (c == crec[li].fail) v = convert_double(tok()[kc[c]]);     // rethrows the conversion error where the reader would have met it
        else v = cval[li * (size_t)nkc + c];
        Xraw[(size_t)(1 + c) * N + i] = v;
        if (v == MISSING) { in_cov[i] = 0; break; }
      }
    }
    mark("covariates: parsed");
    if (std::find(is_cat.begin(), is_cat.end(), (uint8_t)1) != is_cat.end()) {
      // dummy variables (Pheno.cpp:716-783, check_categories :985-1011, get_dummies): level 0 goes to the intercept
      std::vector<double> full(Xraw.begin(), Xraw.begin() + N);
      int nfull = 1;
      for (size_t c = 0; c < kc.size(); ++c) {
        double* col = Xraw.data() + (size_t)(1 + c) * N;
        for (int64_t i = 0; i < N; ++i) col[i] *= in_cov[i];
        if (!is_cat[c]) { full.insert(full.end(), col, col + N); ++nfull; continue; }
        const int nlev = (int)levels[c].size();
        if (nlev > p.max_cat_levels)
          throw std::runtime_error("too many categories for covariate: " + cov_names[c] + " (=" + std::to_string(nlev) + "). Either use '--maxCatLevels' or combine categories.");
        if (nlev == 1) sout << "WARNING: covariate ' " << cov_names[c] << "' only has a single category so it will be ignored\n";
        int top = 0;
        for (int64_t i = 0; i < N; ++i) top = std::max(top, (int)col[i]);
        for (int lvl = 1; lvl <= top; ++lvl) {
          for (int64_t i = 0; i < N; ++i) full.push_back(col[i] == lvl ? 1.0 : 0.0);
          ++nfull;
        }
      }
      Xraw.swap(full);
      ncols = nfull;
    }
    int64_t nc = 0;
    for (int64_t i = 0; i < N; ++i) nc += in_cov[i];
    if (nc == 0) throw std::runtime_error("none of the individuals have covariate data (check sample IDs across files)");
    sout << "   -number of individuals with covariate data = " << nc << "\n";
  } else {
    Xraw.assign((size_t)N, 1.0);
  }
  // masks (Pheno.cpp:101, :810-841)
  const bool strict = p.strict || (r.P == 1 && !p.t2e);
  r.ain.assign(N, 0);
  r.n_analyzed = 0;
  for (int64_t i = 0; i < N; ++i) {
    bool any = false, all = true;
    for (int q = 0; q < r.P; ++q) { any |= r.mask[(size_t)q * N + i] != 0; all &= r.mask[(size_t)q * N + i] != 0; }
    r.ain[i] = (in_pheno[i] && in_cov[i] && (strict ? all : any)) ? 1 : 0;
    r.n_analyzed += r.ain[i];
  }
  if (r.n_analyzed < 1) throw std::runtime_error("sample size cannot be < 1.");
  sout << " * number of individuals used in analysis = " << r.n_analyzed << "\n";
  if (ncols >= N) throw std::runtime_error("Number of covariates is greater than sample size!");
  r.neff.assign(r.P, 0.0);
  for (int q = 0; q < r.P; ++q)
    for (int64_t i = 0; i < N; ++i) {
      r.mask[(size_t)q * N + i] &= r.ain[i];
      r.Y[(size_t)q * N + i] *= r.ain[i];
      if (p.bt || p.ct || p.t2e) r.Yraw[(size_t)q * N + i] *= r.ain[i];
      if (p.t2e) r.Yevent[(size_t)q * N + i] *= r.ain[i];
      r.neff[q] += r.mask[(size_t)q * N + i];
    }
  for (int c = 0; c < ncols; ++c)
    for (int64_t i = 0; i < N; ++i) Xraw[(size_t)c * N + i] *= (r.ain[i] && in_cov[i]) ? 1.0 : 0.0;
  if (p.rint) {  // apply_rint / rint_pheno (Pheno.cpp:111-115, :1937-2010): ranks with ties averaged -> normal quantiles
    sout << "   -applying RINT to all phenotypes\n";
    for (int q = 0; q < r.P; ++q) {
      std::vector<std::pair<double, int64_t>> yv;
      for (int64_t i = 0; i < N; ++i)
        if (r.Y[(size_t)q * N + i] != MISSING && r.mask[(size_t)q * N + i]) yv.emplace_back(r.Y[(size_t)q * N + i], i);
      std::stable_sort(yv.begin(), yv.end(), [](const std::pair<double, int64_t>& a, const std::pair<double, int64_t>& b) { return a.first < b.first; });
      const size_t nv = yv.size();
      for (size_t a = 0; a < nv;) {
        size_t b = a + 1;
        while (b < nv && yv[b].first == yv[a].first) ++b;
        const double rank = (double)(a + 1) + (double)(b - a - 1) / 2.0;
        for (size_t k = a; k < b; ++k)
          r.Y[(size_t)q * N + yv[k].second] = norm_quantile((rank - 3.0 / 8.0) / ((double)nv - 2.0 * (3.0 / 8.0) + 1.0));
        a = b;
      }
    }
  }
  // pheno_impute_miss (QT): missing -> mean over analysed non-missing, then mask
  for (int q = 0; q < r.P && (p.bt || p.ct || p.t2e); ++q) {  // non-QT: mean over the unmasked entries (Pheno.cpp:1921-1930)
    double total = 0.0, ns = 0.0;
    for (int64_t i = 0; i < N; ++i) if (r.mask[(size_t)q * N + i]) { total += r.Y[(size_t)q * N + i]; ns += 1.0; }
    for (int64_t i = 0; i < N; ++i) {
      double& v = r.Y[(size_t)q * N + i];
      if (!r.mask[(size_t)q * N + i]) v = total / ns;
      v *= r.mask[(size_t)q * N + i];
    }
  }
  for (int q = 0; q < r.P && !(p.bt || p.ct || p.t2e); ++q) {
    double total = 0.0, ns = 0.0;
    std::set<double> distinct;
    for (int64_t i = 0; i < N; ++i) {
      const double v = r.Y[(size_t)q * N + i];
      if (v != MISSING) { total += v; if (r.ain[i]) { ns += 1.0; if (distinct.size() < 3) distinct.insert(v); } }
    }
    if (distinct.size() <= 2 && !p.force_qt)  // Pheno.cpp:907-925
      throw std::runtime_error("phenotype '" + r.pheno_names[q] + "' has very few unique values (=" + std::to_string(distinct.size()) + "). If you really want to analyze it as a QT, use --force-qt.");
    for (int64_t i = 0; i < N; ++i) {
      double& v = r.Y[(size_t)q * N + i];
      if (v == MISSING) v = total / ns;
      v *= r.mask[(size_t)q * N + i];
    }
  }
  if (p.t2e) {   // prep_run (Pheno.cpp:1078-1103) + getBasis with trait_mode 3 (:1663-1667): constant columns (the intercept) are dropped, the
                 // others centred -- every row, analysed or not, as the reference does -- and scaled by their sd over the analysed samples
    std::vector<double> X2;
    int kept = 0;
    for (int c = 0; c < ncols; ++c) {
      double mu = 0.0, ss = 0.0;
      for (int64_t i = 0; i < N; ++i) mu += Xraw[(size_t)c * N + i];
      mu /= (double)N;
      for (int64_t i = 0; i < N; ++i) { const double dlt = Xraw[(size_t)c * N + i] - mu; ss += dlt * dlt; }
      const double sd = std::sqrt(ss) / std::sqrt((double)r.n_analyzed);
      if (!(sd > 1e-6)) continue;        // const_cov_cox_tol, Regenie.hpp:228
      for (int64_t i = 0; i < N; ++i) X2.push_back((Xraw[(size_t)c * N + i] - mu) / sd);
      ++kept;
    }
    if (kept == 0) throw std::runtime_error("--t2e without a non-constant covariate is not built (the null Cox model needs one).");
    Xraw.swap(X2);
    ncols = kept;
  }
  mark("masks / imputation / covariate prep");
  // getBasis (Pheno.cpp:1660-1681)
  std::vector<double> xtx((size_t)ncols * ncols, 0.0), d, V;
  for (int a = 0; a < ncols; ++a)
    for (int b = a; b < ncols; ++b) {
      double s = 0.0;
      for (int64_t i = 0; i < N; ++i) s += Xraw[(size_t)a * N + i] * Xraw[(size_t)b * N + i];
      xtx[(size_t)a * ncols + b] = xtx[(size_t)b * ncols + a] = s;
    }
  jacobi_eigh(xtx, ncols, d, V);
  int nz = 0;
  for (int j = 0; j < ncols; ++j) nz += d[j] > d[ncols - 1] * 1e-15;
  r.C = nz;
  r.X.assign((size_t)N * nz, 0.0);
  for (int j = 0; j < nz; ++j) {
    const int src = ncols - nz + j;
    const double inv = 1.0 / std::sqrt(d[src]);
    for (int c = 0; c < ncols; ++c) {
      const double v = V[(size_t)c * ncols + src] * inv;
      for (int64_t i = 0; i < N; ++i) r.X[(size_t)j * N + i] += Xraw[(size_t)c * N + i] * v;
    }
  }
  // fit_null_logistic (Step1_Models.cpp:54-154): offsets of the covariate-only logistic model
  r.pheno_pass.assign(r.P, 1);
  if (p.bt) {
    sout << "   -fitting null logistic regression on binary phenotypes...";
    r.offset.assign((size_t)N * r.P, 0.0);
    for (int q = 0; q < r.P; ++q) {
      std::vector<double> eta;
      std::vector<double> b0;
      LogisticState lst;      // the second attempt goes on from the first one's state, as the reference's in-place arguments make it
      bool ok = fit_logistic(r.Yraw.data() + (size_t)q * N, r.X.data(), r.mask.data() + (size_t)q * N, N, nz, p, true, eta, nullptr, nullptr, &b0, &lst);
      if (!ok) ok = fit_logistic(r.Yraw.data() + (size_t)q * N, r.X.data(), r.mask.data() + (size_t)q * N, N, nz, p, false, eta, nullptr, nullptr, &b0, &lst);
      if (!ok) { r.pheno_pass[q] = 0; continue; }
      if (p.write_null_firth) { r.bhat_start.resize((size_t)r.P * nz, 0.0); std::copy(b0.begin(), b0.end(), r.bhat_start.begin() + (size_t)q * nz); }   // Step1_Models.cpp:138
      for (int64_t i = 0; i < N; ++i) r.offset[(size_t)q * N + i] = eta[i];
    }
    sout << "done\n";
  } else if (p.t2e) {   // fit_null_cox in step 1 (Step1_Models.cpp:353-440): the covariates' linear predictor is the level-1 offset
    sout << "   -fitting null cox regression on time-to-event phenotypes...";
    r.offset.assign((size_t)N * r.P, 0.0);
    for (int q = 0; q < r.P; ++q) {
      std::vector<double> eta;
      const double *tq = r.Yraw.data() + (size_t)q * N, *eq = r.Yevent.data() + (size_t)q * N;
      const uint8_t* mq = r.mask.data() + (size_t)q * N;
      // Step1_Models.cpp:415-436: coordinate descent, then the Newton solver with and without its step-halving tolerance; a trait
      // that none of them fits is dropped with a warning (pheno_pass = false) and the run goes on with the others
      bool ok = cox_null_fit(tq, eq, mq, r.X.data(), N, nz, p, eta) && !getenv("RG_COX_NULL_FORCE_NEWTON");
      if (!ok) ok = cox_null_newton(tq, eq, mq, r.X.data(), N, nz, p, 2.5e-4, eta);
      if (!ok) ok = cox_null_newton(tq, eq, mq, r.X.data(), N, nz, p, 0.0, eta);
      if (!ok) {
        r.pheno_pass[q] = 0;
        sout << "\n     WARNING: step1 cox null regression did not converge for phenotype '" << r.pheno_names[q] << "'.";
        continue;
      }
      for (int64_t i = 0; i < N; ++i) r.offset[(size_t)q * N + i] = eta[i];
    }
    sout << "done\n";
  } else if (p.ct) {
    sout << "   -fitting null poisson regression...";
    r.offset.assign((size_t)N * r.P, 0.0);
    for (int q = 0; q < r.P; ++q) {
      std::vector<double> eta;
      if (!fit_poisson(r.Yraw.data() + (size_t)q * N, r.X.data(), r.mask.data() + (size_t)q * N, N, nz, p, eta)) {
        r.pheno_pass[q] = 0;
        sout << "\n     WARNING: poisson regression did not converge for phenotype '" << r.pheno_names[q] << "'.";
        continue;
      }
      for (int64_t i = 0; i < N; ++i) r.offset[(size_t)q * N + i] = eta[i];
    }
    sout << "done\n";
  }
  mark("basis + null models");
  // residualize_phenotypes (Pheno.cpp:1799-1834)
  sout << "   -residualizing and scaling phenotypes...";
  r.scale_Y.assign(r.P, 1.0);
  for (int q = 0; q < r.P; ++q) {
    std::vector<double> beta(nz, 0.0);
    for (int j = 0; j < nz; ++j)
      for (int64_t i = 0; i < N; ++i) beta[j] += r.Y[(size_t)q * N + i] * r.X[(size_t)j * N + i];
    double ss = 0.0;
    for (int64_t i = 0; i < N; ++i) {
      double fit = 0.0;
      for (int j = 0; j < nz; ++j) fit += r.X[(size_t)j * N + i] * beta[j];
      double& y = r.Y[(size_t)q * N + i];
      y -= fit * r.mask[(size_t)q * N + i];
      ss += y * y;
    }
    r.scale_Y[q] = std::sqrt(ss) / std::sqrt(r.neff[q] - nz);
    if (!r.pheno_pass[q]) r.scale_Y[q] = 1.0;
    if (r.scale_Y[q] < 1e-6) throw std::runtime_error("phenotype '" + r.pheno_names[q] + "' has sd=0.");
    for (int64_t i = 0; i < N; ++i) r.Y[(size_t)q * N + i] /= r.scale_Y[q];
  }
  mark("residualize phenotypes");
  sout << "done\n";
}

// prep_parallel_l0 (Data.cpp:818-859): header + line `job_num` of the master file
void prep_parallel_l0(Run& r) {
  const Params& p = r.p;
  sout << " * running jobs in parallel (job #" << p.job_num << ")\n";
  std::ifstream f(p.split_file);
  if (!f) throw std::runtime_error("cannot open file : " + p.split_file);
  std::string line;
  if (!std::getline(f, line)) throw std::runtime_error("cannot read header line in master file.");
  long long ng = 0; int bsz = 0;
  if (sscanf(line.c_str(), "%lld %d", &ng, &bsz) != 2 || bsz != p.bsize) throw std::runtime_error("invalid header line in master file.");
  r.parallel_nGeno = ng;
  for (int k = 1; k <= p.job_num; ++k)
    if (!std::getline(f, line)) throw std::runtime_error("could not read line " + std::to_string(p.job_num + 1) + " (check number of lines in file).");
  char pref[4096];
  if (sscanf(line.c_str(), "%4095s %d %d", pref, &r.parallel_nBlocks, &r.parallel_nSnps) != 3)
    throw std::runtime_error("could not read line " + std::to_string(p.job_num + 1) + " (check number of lines and format in file).");
  r.job_prefix = pref;
}

// prep_parallel_l1 (Data.cpp:862-908)
void prep_parallel_l1(Run& r, int total_n_block, int64_t n_variants) {
  const Params& p = r.p;
  std::ifstream f(p.split_file);
  if (!f) throw std::runtime_error("cannot open file : " + p.split_file);
  std::string line;
  if (!std::getline(f, line)) throw std::runtime_error("cannot read header line in master file.");
  long long ng = 0; int bsz = 0;
  if (sscanf(line.c_str(), "%lld %d", &ng, &bsz) != 2 || bsz != p.bsize) throw std::runtime_error("invalid header line in master file.");
  r.parallel_nGeno = ng;
  int nblocks = 0, lineread = 0;
  int64_t nsnps = 0;
  while (std::getline(f, line)) {
    char pref[4096]; int nb = 0, ns = 0;
    if (sscanf(line.c_str(), "%4095s %d %d", pref, &nb, &ns) != 3)
      throw std::runtime_error("could not read line " + std::to_string(lineread + 2) + " (check number of lines and format in file).");
    r.bstart.push_back(nblocks); r.btot.push_back(nb); r.mprefix.push_back(pref);
    if (nblocks > total_n_block || nb < 0) throw std::runtime_error("invalid block information in master file at line " + std::to_string(lineread + 2) + ".");
    nblocks += nb; nsnps += ns; ++lineread;
  }
  if (nblocks != total_n_block || nsnps != n_variants)
    throw std::runtime_error("number of blocks/variants in master file '" + p.split_file + "' doesn't match that in the analysis.");
  sout << " * using results from running " << lineread << " parallel jobs at level 0\n";
}

}  // namespace rgdrv
