// regenie-amd: C++ host driver for `--step 1` on MI355X.
//
// Keeps the reference's command-line surface for Step 1 (src/Regenie.cpp:146-371 option table, subset
// listed in SURVEY.md section 2 row 1) and its outputs (<out>.log, <out>_pred.list, <out>_<k>.loco,
// optional <out>_<k>.prs / <out>_prs.list; src/Data.cpp:956-1129, :1795-1975) and calls the HIP library
// through the C ABI of include/rg_step1.h for everything Data::level_0_calculations, ridge_level_1 and
// make_predictions do.  Host-side prerequisites (text parsing, masks, covariate basis, phenotype
// residualisation, fold / block bookkeeping, LOCO assembly, writers) follow the reference functions
// cited next to each routine.  There is no CPU compute fallback for the hot path.
//
// Served: --qt / --bt, K-fold CV / --loocv (and the reference's automatic LOOCV for --bt below 5,000 samples), and the
// file protocol of the level-0 job split: --split-l0 PFX,N / --run-l0 PFX.master,k / --run-l1 PFX.master [--keep-l0]
// (src/Data.cpp:232-309, :818-908; raw double N x (blocks*R0) files of Step1_Models.cpp:728-734).
// Genotype input: --bed PFX (bed/bim/fam), --pgen PFX (pgen/pvar/psam: hardcalls decoded to the same 2-bit rows, dosage files
// to doubles for the fp64 level 0; include/rg_pgen.h) or --bgen FILE [--sample FILE] (BGEN v1.2, 8-bit; include/rg_bgen.h);
// gzipped text inputs and --gz outputs through zlib (Files.cpp:38-160).  Not served (explicit errors, never silent): BGEN files
// other than layout 2 with 8-bit probabilities.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <fcntl.h>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <charconv>
#include <climits>
#include <unistd.h>
#include <zlib.h>

#include "../../include/rg_bgen.h"
#include "../../include/rg_pgen.h"
#include "../../include/rg_step1.h"
#include "../../include/rg_step2.h"

namespace {

const double MISSING = -999.0;  // Regenie.hpp:215

struct Params {
  int step = 0;
  std::string bed, pgen, bgen, sample_file, pheno_file, covar_file, out = "regenie_out";
  std::vector<std::string> keep, remove, extract, exclude, pheno_cols, covar_cols, cat_covar;
  int max_cat_levels = 10;
  bool rint = false;
  int bsize = 0, cv_folds = 5, n_ridge_l0 = 5, n_ridge_l1 = 5, nchrom = 23, threads = 0;
  bool firth = false, firth_approx = false, firth_se = false;   // --firth --approx [--firth-se] (step 2, binary traits)
  bool write_null_firth = false;          // --write-null-firth (step 1 or 2): the null approximate-Firth estimates per chromosome, PFX_<k>.firth + PFX_firth.list
  std::string use_null_firth;             // --use-null-firth LIST (step 2): start values of the null Firth fits
  bool spa = false;                                             // --spa (step 2, binary traits)
  double pthresh = 0.05;                                        // --pThresh: score tests below it get the correction
  double min_info = 0.0; bool set_min_info = false;             // --minINFO (step 2, dosages)
  bool bt = false, ct = false, loocv = false, strict = false, ref_first = false, use_rel_path = false,
       print_prs = false, force_step1 = false, lowmem = false, force_qt = false, cc12 = false, gz = false;
  bool t2e = false;                        // --t2e: time-to-event traits (step 1: Cox ridge at level 1)
  std::vector<std::string> event_cols;     // --eventColList, matching --phenoColList (the time columns) in order
  int min_case_count = 10, niter_max = 50, niter_max_line_search = 25, niter_max_ridge = 100;
  // level-0 job split (Data.cpp:232-309, :818-908)
  std::string split_file;              // --split-l0 prefix / --run-l0, --run-l1 master file
  int njobs = 0, job_num = 0;
  bool split_l0 = false, run_l0 = false, run_l1 = false, keep_l0 = false;
  std::vector<double> setl0, setl1;
  int device = 0;
  // one node, several GPUs (no counterpart option in the reference, whose job split goes through files): --gpus N deals the
  // SNP blocks to N GPUs like write_l0_master (Data.cpp:270-302), one host thread per GPU, and replaces the job files by one
  // exchange of the level-0 predictors (rg_l0_finish).  --transport rccl (default) | peer; --single-device puts every rank on
  // --device (test mode, needs --transport peer); --force-collectives runs the exchange code with a single GPU as well.
  int gpus = 1;
  int transport = RG_TRANSPORT_RCCL;
  bool single_device = false, force_collectives = false;
  bool l1_shared = false;   // --l1-shared: all-gather + shared level 1 even when every GPU could own a phenotype
  // --step 2 (single-variant association test, quantitative traits: Data::test_snps_fast, Data.cpp:2230-2360)
  std::string pred_list;    // --pred: the _pred.list of step 1
  double min_mac = 5;       // --minMAC (Regenie.hpp:311)
};

// a worker thread of a multi-GPU step-2 run logs into its own buffer (tl_log): the parts' logs are appended in order afterwards
thread_local std::ostringstream* tl_log = nullptr;
struct Log {  // mstream (Regenie.hpp:120-142): tee to stdout and <out>.log
  std::ofstream f;
  template <class T>
  Log& operator<<(const T& v) { if (tl_log) { *tl_log << v; return *this; } std::cout << v; if (f.is_open()) f << v; return *this; }
  Log& operator<<(std::ostream& (*m)(std::ostream&)) { if (tl_log) { *tl_log << m; return *this; } std::cout << m; if (f.is_open()) f << m; return *this; }
};
Log sout;
std::mutex g_reader_mu;      // the .pgen / .bgen readers keep per-handle state: one block read at a time (multi-GPU step 2)

// Files (Files.cpp:38-160): a file whose name ends in ".gz" and starts with the gzip magic is read through zlib, anything
// else as plain text; `--gz` writes the .loco / .prs outputs through zlib.  (The reference needs Boost Iostreams for this.)
bool ends_with_gz(const std::string& fn) { return fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".gz") == 0; }
bool file_exists(const std::string& fn) { return access(fn.c_str(), F_OK) == 0; }

class GzInBuf : public std::streambuf {
 public:
  explicit GzInBuf(const std::string& fn) : f_(gzopen(fn.c_str(), "rb")), buf_(1 << 16) {}
  ~GzInBuf() override { if (f_) gzclose(f_); }
  bool ok() const { return f_ != nullptr; }
 protected:
  int_type underflow() override {
    if (gptr() < egptr()) return traits_type::to_int_type(*gptr());
    const int n = f_ ? gzread(f_, buf_.data(), (unsigned)buf_.size()) : 0;
    if (n <= 0) return traits_type::eof();
    setg(buf_.data(), buf_.data(), buf_.data() + n);
    return traits_type::to_int_type(*gptr());
  }
 private:
  gzFile f_;
  std::vector<char> buf_;
};

class TextIn : public std::istream {  // Files::openForRead
 public:
  explicit TextIn(const std::string& fn) : std::istream(nullptr) {
    bool gz = false;
    if (ends_with_gz(fn)) {  // isGzipped(filename, true): extension first, then the two magic bytes
      std::ifstream t(fn, std::ios::binary);
      unsigned char h[2] = {0, 0};
      t.read((char*)h, 2);
      gz = t && h[0] == 0x1f && h[1] == 0x8b;
    }
    if (gz) {
      gz_.reset(new GzInBuf(fn));
      if (gz_->ok()) rdbuf(gz_.get()); else setstate(std::ios::failbit);
    } else {
      if (plain_.open(fn, std::ios::in)) rdbuf(&plain_); else setstate(std::ios::failbit);
    }
  }
 private:
  std::filebuf plain_;
  std::unique_ptr<GzInBuf> gz_;
};

class GzOutBuf : public std::streambuf {
 public:
  explicit GzOutBuf(const std::string& fn) : f_(gzopen(fn.c_str(), "wb")) {}
  ~GzOutBuf() override { if (f_) gzclose(f_); }
  bool ok() const { return f_ != nullptr; }
 protected:
  int_type overflow(int_type c) override {
    if (c == traits_type::eof()) return traits_type::not_eof(c);
    const char ch = traits_type::to_char_type(c);
    return (f_ && gzwrite(f_, &ch, 1) == 1) ? c : traits_type::eof();
  }
  std::streamsize xsputn(const char* p, std::streamsize n) override {
    if (!f_ || n <= 0) return 0;
    return gzwrite(f_, p, (unsigned)n) > 0 ? n : 0;
  }
 private:
  gzFile f_;
};

class TextOut : public std::ostream {  // Files::openForWrite
 public:
  TextOut(const std::string& fn, bool gz) : std::ostream(nullptr) {
    if (gz) {
      gz_.reset(new GzOutBuf(fn));
      if (gz_->ok()) rdbuf(gz_.get()); else setstate(std::ios::failbit);
    } else {
      if (plain_.open(fn, std::ios::out)) rdbuf(&plain_); else setstate(std::ios::failbit);
    }
  }
 private:
  std::filebuf plain_;
  std::unique_ptr<GzOutBuf> gz_;
};

std::vector<std::string> split_ws(const std::string& s) {     // the tokens `is >> t` would give (a stream per line cost 2 s of a 500,000-sample run)
  std::vector<std::string> out;
  const size_t n = s.size();
  size_t i = 0;
  while (i < n) {
    while (i < n && std::isspace((unsigned char)s[i])) ++i;
    size_t j = i;
    while (j < n && !std::isspace((unsigned char)s[j])) ++j;
    if (j > i) out.emplace_back(s, i, j - i);
    i = j;
  }
  return out;
}
std::vector<std::string> split_char(const std::string& s, char c) {
  std::vector<std::string> out;
  std::string t;
  std::istringstream is(s);
  while (std::getline(is, t, c)) if (!t.empty()) out.push_back(t);
  return out;
}

int chr_str_to_int(std::string s, int nchrom) {  // Regenie.cpp:1583-1594
  if (s.compare(0, 3, "chr") == 0) s = s.substr(3);
  if (!s.empty() && isdigit((unsigned char)s[0])) {
    int c = atoi(s.c_str());
    if (c >= 1 && c <= nchrom) return c;
  } else if (s == "X" || s == "XY" || s == "Y" || s == "PAR1" || s == "PAR2") return nchrom;
  return -1;
}

double convert_double(const std::string& v) {  // Regenie.cpp:1663-1675
  if (v == "NA" || v == "nan" || v == "inf") return MISSING;
  char* end = nullptr;
  double d = strtod(v.c_str(), &end);
  if (end == v.c_str()) throw std::runtime_error("could not convert value to double: '" + v + "'");
  return d;
}

std::string cpp_double(double v) {  // default ostream formatting (precision 6)
  std::ostringstream o;
  o << v;
  return o.str();
}

std::set<std::string> read_id_files(const std::vector<std::string>& files) {  // Geno.cpp:1382-1441
  std::set<std::string> ids;
  for (auto& fn : files) {
    TextIn f(fn);
    if (!f) throw std::runtime_error("cannot open file : " + fn);
    std::string line;
    while (std::getline(f, line)) {
      auto t = split_ws(line);
      if (t.size() < 2) throw std::runtime_error("incorrectly formatted file: " + fn);
      ids.insert(t[0] + "_" + t[1]);
    }
  }
  return ids;
}
std::set<std::string> read_snp_files(const std::vector<std::string>& files) {
  std::set<std::string> ids;
  for (auto& fn : files) {
    TextIn f(fn);
    if (!f) throw std::runtime_error("cannot open file : " + fn);
    std::string line;
    while (std::getline(f, line)) {
      auto t = split_ws(line);
      if (!t.empty()) ids.insert(t[0]);
    }
  }
  return ids;
}

// symmetric eigen-decomposition (cyclic Jacobi), ascending eigenvalues; n is the covariate count
void jacobi_eigh(std::vector<double> A, int n, std::vector<double>& d, std::vector<double>& V) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)p * n + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - s * akq;
          A[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - s * aqk;
          A[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - s * vkq;
          V[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  d.resize(n);
  for (int i = 0; i < n; ++i) d[i] = A[(size_t)i * n + i];
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] < d[b]; });
  std::vector<double> d2(n), V2((size_t)n * n);
  for (int j = 0; j < n; ++j) {
    d2[j] = d[idx[j]];
    for (int k = 0; k < n; ++k) V2[(size_t)k * n + j] = V[(size_t)k * n + idx[j]];
  }
  d.swap(d2);
  V.swap(V2);
}

[[noreturn]] void usage_error(const std::string& m) { throw std::runtime_error(m); }

Params parse_args(int argc, char** argv) {
  Params p;
  auto need = [&](int& i) -> std::string {
    if (i + 1 >= argc) usage_error(std::string("option '") + argv[i] + "' needs a value");
    return argv[++i];
  };
  auto list = [&](std::vector<std::string>& dst, const std::string& v) {
    for (auto& s : split_char(v, ',')) dst.push_back(s);
  };
  bool saw_pheno_col = false, saw_pheno_collist = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--step") p.step = atoi(need(i).c_str());
    else if (a == "--bed") p.bed = need(i);
    else if (a == "--phenoFile" || a == "--p") p.pheno_file = need(i);
    else if (a == "--covarFile" || a == "--c") p.covar_file = need(i);
    else if (a == "--phenoCol" || a == "--phenoColList") { if (a == "--phenoCol") saw_pheno_col = true; else saw_pheno_collist = true; list(p.pheno_cols, need(i)); }
    else if (a == "--eventColList") list(p.event_cols, need(i));
    else if (a == "--t2e") { p.t2e = true; p.bt = p.ct = false; }
    else if (a == "--t2e-event-l0" || a == "--t2e-l1-pi6") usage_error("option '" + a + "' is not built (the level-0 response of a time-to-event trait is its time column, the penalties come from the score at beta = 0).");
    else if (a == "--covarCol" || a == "--covarColList") list(p.covar_cols, need(i));
    else if (a == "--catCovarList") list(p.cat_covar, need(i));
    else if (a == "--maxCatLevels") p.max_cat_levels = atoi(need(i).c_str());
    else if (a == "--apply-rint") p.rint = true;
    else if (a == "--keep") list(p.keep, need(i));
    else if (a == "--remove") list(p.remove, need(i));
    else if (a == "--extract") list(p.extract, need(i));
    else if (a == "--exclude") list(p.exclude, need(i));
    else if (a == "--bsize" || a == "--b") p.bsize = atoi(need(i).c_str());
    else if (a == "--cv") p.cv_folds = atoi(need(i).c_str());
    else if (a == "--l0") p.n_ridge_l0 = atoi(need(i).c_str());
    else if (a == "--l1") p.n_ridge_l1 = atoi(need(i).c_str());
    else if (a == "--setl0") { for (auto& s : split_char(need(i), ',')) p.setl0.push_back(atof(s.c_str())); }
    else if (a == "--setl1") { for (auto& s : split_char(need(i), ',')) p.setl1.push_back(atof(s.c_str())); }
    else if (a == "--out" || a == "--o") p.out = need(i);
    else if (a == "--threads") p.threads = atoi(need(i).c_str());
    else if (a == "--nauto") p.nchrom = atoi(need(i).c_str()) + 1;
    else if (a == "--device") p.device = atoi(need(i).c_str());
    else if (a == "--lowmem-prefix") { need(i); p.lowmem = true; }
    else if (a == "--qt") { p.bt = false; p.ct = false; p.t2e = false; }
    else if (a == "--bt") { p.bt = true; p.ct = false; p.t2e = false; }
    else if (a == "--ct") { p.ct = true; p.bt = false; p.t2e = false; }
    else if (a == "--loocv") p.loocv = true;
    else if (a == "--strict") p.strict = true;
    else if (a == "--ref-first") p.ref_first = true;
    else if (a == "--use-relative-path") p.use_rel_path = true;
    else if (a == "--print-prs") p.print_prs = true;
    else if (a == "--force-step1") p.force_step1 = true;
    else if (a == "--force-qt") p.force_qt = true;
    else if (a == "--lowmem") p.lowmem = true;
    else if (a == "--1" || a == "--cc12") p.cc12 = true;
    else if (a == "--minCaseCount") p.min_case_count = atoi(need(i).c_str());
    else if (a == "--niter") { p.niter_max = atoi(need(i).c_str()); p.niter_max_ridge = p.niter_max; }  // Regenie.cpp:483
    else if (a == "--gz") p.gz = true;
    else if (a == "--pgen") p.pgen = need(i);
    else if (a == "--bgen") p.bgen = need(i);
    else if (a == "--sample") p.sample_file = need(i);
    else if (a == "--split-l0") {
      auto t = split_char(need(i), ',');
      if (t.size() != 2) usage_error("must specify number of jobs for --split-l0 (i.e. prefix,njobs).");
      p.split_l0 = true; p.split_file = t[0]; p.njobs = atoi(t[1].c_str());
    } else if (a == "--run-l0") {
      auto t = split_char(need(i), ',');
      if (t.size() != 2) usage_error("must specify job number for --run-l0 (i.e. master_file,job_number).");
      p.run_l0 = true; p.split_file = t[0]; p.job_num = atoi(t[1].c_str());
      if (p.job_num < 1) usage_error("invalid job number for --run-l0 (must be >=1).");
    } else if (a == "--run-l1") { p.run_l1 = true; p.split_file = need(i); }
    else if (a == "--keep-l0") p.keep_l0 = true;
    else if (a == "--gpus") p.gpus = atoi(need(i).c_str());
    else if (a == "--transport") {
      const std::string t = need(i);
      if (t == "rccl") p.transport = RG_TRANSPORT_RCCL;
      else if (t == "peer") p.transport = RG_TRANSPORT_PEER;
      else usage_error("--transport must be rccl or peer");
    }
    else if (a == "--single-device") p.single_device = true;
    else if (a == "--force-collectives") p.force_collectives = true;
    else if (a == "--l1-shared") p.l1_shared = true;
    else if (a == "--pred") p.pred_list = need(i);
    else if (a == "--minMAC") p.min_mac = atof(need(i).c_str());
    else if (a == "--minINFO") { p.min_info = atof(need(i).c_str()); p.set_min_info = true; }
    else if (a == "--firth") p.firth = true;
    else if (a == "--approx") p.firth_approx = true;
    else if (a == "--firth-se") p.firth_se = true;
    else if (a == "--write-null-firth") p.write_null_firth = true;
    else if (a == "--use-null-firth") p.use_null_firth = need(i);
    else if (a == "--pThresh") p.pthresh = atof(need(i).c_str());
    else if (a == "--spa") p.spa = true;
    else usage_error("unrecognised option '" + a + "'");
  }
  if (p.bt) p.rint = false;  // Regenie.cpp:432
  if (p.step != 1 && p.step != 2) usage_error("specify which mode regenie should be running using option --step.");
  // time-to-event traits (Regenie.cpp:570-587, :1197-1201)
  if (p.t2e && !p.event_cols.empty() && saw_pheno_col) usage_error("You must specify TTE phenotypes using '--phenoColList' (matching in order with events in '--eventColList').");
  if (p.t2e && (p.event_cols.empty() || !saw_pheno_collist)) usage_error("You must specify both '--phenoColList' and '--eventColList' (same order) for time-to-event analysis.");
  if (!p.event_cols.empty() && !p.t2e) usage_error("Option --eventColList must be used with '--t2e' for time-to-event analysis");
  if (p.t2e && p.event_cols.size() != p.pheno_cols.size()) usage_error("'--phenoColList' and '--eventColList' must name the same number of columns.");
  if (p.t2e && p.step == 2) usage_error("--step 2 --t2e (the Cox score test) is not built: time-to-event traits are supported in step 1.");
  if (p.t2e && (p.run_l0 || p.run_l1 || p.split_l0)) usage_error("--t2e with the --split-l0 / --run-l0 / --run-l1 file protocol is not built.");
  if (p.t2e && p.loocv) { std::cout << "WARNING: option --loocv cannot be used with option --t2e.\n"; p.loocv = false; }
  if (p.t2e) p.rint = false;
  if (p.step == 2) {
    if (p.pred_list.empty()) usage_error("option '--pred' is required (use the _pred.list file written by step 1).");
    if (p.firth && !p.bt) usage_error("option '--firth' applies to binary traits (--bt).");
    if (p.spa && !p.bt) usage_error("option '--spa' applies to binary traits (--bt).");
    if (p.spa && p.firth) usage_error("cannot use both '--firth' and '--spa'.");
    if (p.spa && !(p.pthresh > 0 && p.pthresh < 1)) usage_error("'--pThresh' must be in (0,1).");
    if (p.firth && !(p.pthresh > 0 && p.pthresh < 1)) usage_error("'--pThresh' must be in (0,1).");
    if (p.min_mac < 0.5) usage_error("minimum MAC must be at least 0.5.");   // Regenie.cpp:1054
    if (p.set_min_info && (p.min_info < 0 || p.min_info > 1)) usage_error("minimum info score must be in [0,1].");
    if (p.force_collectives) usage_error("--force-collectives applies to step 1 (step 2 has no collective: its blocks are independent).");
  }
  if (!p.use_null_firth.empty() && !(p.step == 2 && p.firth && p.firth_approx)) usage_error("option --use-null-firth only wors with approximate Firth test.");   // Regenie.cpp:1216-1217
  if (p.write_null_firth && ((p.step == 2 && !(p.firth && p.firth_approx)) || (p.step == 1 && !p.bt))) {   // Regenie.cpp:1218-1222
    std::cout << "WARNING: option --write-null-firth only works for BTs with approximate Firth test.\n";
    p.write_null_firth = false;
  }
  if ((int)!p.bed.empty() + (int)!p.pgen.empty() + (int)!p.bgen.empty() != 1) usage_error("must use either --bed,--bgen or --pgen.");  // Regenie.cpp:419-420
  if (p.pheno_file.empty()) usage_error("option '--phenoFile' is required.");
  if (p.bsize < 1) usage_error("must specify the block size using '--bsize'.");
  if (p.cv_folds < 2) usage_error("number of CV folds must be at least 2");
  if (p.gpus < 1) usage_error("--gpus must be at least 1");
  if (p.step == 1 && p.single_device && p.gpus > 1 && p.transport != RG_TRANSPORT_PEER) usage_error("--single-device needs --transport peer (RCCL refuses two ranks on one device)");
  if ((p.gpus > 1 || p.force_collectives) && (p.run_l0 || p.run_l1 || p.split_l0)) usage_error("--gpus / --force-collectives cannot be combined with the --split-l0 / --run-l0 / --run-l1 file protocol");
  return p;
}

struct Run {
  Params p;
  std::vector<double> bhat_start;        // [P][C] null logistic estimates (--write-null-firth in step 1: start of the null Firth fits)
  // genotype meta
  std::vector<std::string> fam_ids;      // FID_IID, file order
  std::vector<int> snp_chrom;            // kept variants
  std::vector<int64_t> snp_offset;
  std::vector<int> chr_read;             // chromosomes in file order
  std::vector<std::string> snp_ids;      // kept variants
  std::vector<int64_t> snp_pos;          // kept variants: physpos, and the two alleles in output order (Geno.cpp:546-553)
  std::vector<std::string> snp_a0, snp_a1;
  // --step 2: the LOCO files of step 1 (blup_read, Pheno.cpp:1241-1391)
  struct Blup { std::string file; std::vector<int64_t> col_sample; std::vector<int64_t> line_off;
                std::vector<std::string> lines; };     // lines: the chromosome rows of a gzipped file (no seeking there), else empty
  std::vector<Blup> blups;               // per phenotype
  // --run-l0 / --run-l1 (prep_parallel_l0 / prep_parallel_l1)
  int64_t parallel_nGeno = 0;            // global number of variants (lambda uses it, Data.cpp:607)
  int parallel_nBlocks = 0, parallel_nSnps = 0;
  std::string job_prefix;
  std::vector<std::string> mprefix; std::vector<int> bstart, btot;
  int64_t n_file = 0, bpr = 0;
  rg_pgen* pgen = nullptr;               // --pgen: open reader (bed rows come from rg_pgen_read_bed_rows)
  bool dosage_mode = false;              // --pgen with dosage tracks / --bgen: rows of doubles, level 0 from rg_l0_blocks_f64
  bool has_male = false;                 // a sample with sex code 1 in the .fam / .psam (Step 2 on chromosome X needs it)
  rg_bgen* bgenh = nullptr;              // --bgen: open reader
  Run() = default;
  Run(const Run&) = delete;
  ~Run() { if (pgen) rg_pgen_close(pgen); if (bgenh) rg_bgen_close(bgenh); }
  // samples
  std::vector<uint8_t> ind_ignore, ain;  // N_file, N
  std::vector<std::string> ids;          // kept, file order
  int64_t N = 0, n_analyzed = 0;
  // phenotypes / covariates
  std::vector<std::string> pheno_names;
  int P = 0, C = 0;
  std::vector<double> Y, X, neff, scale_Y;  // col-major N x P, N x C
  std::vector<uint8_t> mask;                // col-major N x P
  std::vector<double> Yraw, offset;         // BT: phenotypes_raw (0/1) and offset_nullreg, col-major N x P
  std::vector<uint8_t> pheno_pass;          // BT: null logistic model converged
  // --t2e: the phenotypes of the run are the TIME columns; Yraw = their raw values, Yevent the matching event columns (0 / 1),
  // t2e_num[q] = 1-based position of time column q among the selected (time and event) columns of the file: the reference counts
  // both as phenotypes, and names its outputs by that number (out_<num>.loco)
  std::vector<double> Yevent;
  std::vector<int> t2e_num;
  int outnum(int q) const { return t2e_num.empty() ? q + 1 : t2e_num[q]; }
};

// ---- binary traits: covariate-only logistic regression (Step1_Models.cpp:54-222) ---------------------------
const double NUMTOL = 1e-6;                                   // Regenie.hpp numtol
const double NUMTOL_EPS = 10 * 2.220446049250313e-16;         // Regenie.hpp:225
double get_pvec1(double eta) {                                // Step1_Models.cpp:1799-1806
  double pr = 1.0 - 1.0 / (std::exp(eta) + 1.0);
  if (eta < -30.0) pr = NUMTOL_EPS / (1.0 + NUMTOL_EPS);
  if (eta > 30.0) pr = 1.0 / (1.0 + NUMTOL_EPS);
  return pr;
}
// dense solve by Gaussian elimination with partial pivoting (order = number of covariates)
bool solve_dense(std::vector<double> A, std::vector<double> b, int n, std::vector<double>& x) {
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i) if (std::fabs(A[(size_t)i * n + k]) > std::fabs(A[(size_t)piv * n + k])) piv = i;
    if (A[(size_t)piv * n + k] == 0.0) return false;
    if (piv != k) { for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)piv * n + j]); std::swap(b[k], b[piv]); }
    for (int i = k + 1; i < n; ++i) {
      const double f = A[(size_t)i * n + k] / A[(size_t)k * n + k];
      for (int j = k; j < n; ++j) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
      b[i] -= f * b[k];
    }
  }
  x.assign(n, 0.0);
  for (int i = n - 1; i >= 0; --i) {
    double v = b[i];
    for (int j = i + 1; j < n; ++j) v -= A[(size_t)i * n + j] * x[j];
    x[i] = v / A[(size_t)i * n + i];
  }
  return true;
}
// fit_logistic (Step1_Models.cpp:156-222) for one phenotype; offset may be null (zero); eta_out = offset + X beta on success,
// pv_out (optional) the fitted probabilities
bool fit_logistic(const double* y, const double* X, const uint8_t* mask, int64_t N, int C, const Params& prm,
                  bool check_hs_dev, std::vector<double>& eta, const double* offset = nullptr, std::vector<double>* pv_out = nullptr,
                  std::vector<double>* beta_out = nullptr) {
  std::vector<double> beta(C, 0.0), betanew(C, 0.0), pv(N), w(N);
  auto dev = [&](const std::vector<double>& pp) {
    double t = 0.0;
    for (int64_t i = 0; i < N; ++i) if (mask[i]) t -= (y[i] == 0.0) ? std::log(1.0 - pp[i]) : std::log(pp[i]);
    return 2.0 * t;
  };
  eta.assign(N, 0.0);
  for (int64_t i = 0; i < N; ++i) { eta[i] = offset ? offset[i] : 0.0; pv[i] = get_pvec1(eta[i]); }
  double dev_old = dev(pv), dev_new = dev_old, diff_dev = 0.0;
  int niter = 0;
  bool small_score = false;
  while (true) {
    if (++niter > prm.niter_max) break;
    for (int64_t i = 0; i < N; ++i) { w[i] = mask[i] ? pv[i] * (1.0 - pv[i]) : 1.0; if (w[i] == 0.0) return false; }
    std::vector<double> A((size_t)C * C, 0.0), b(C, 0.0);
    for (int64_t i = 0; i < N; ++i) {
      if (!mask[i]) continue;
      const double z = eta[i] - (offset ? offset[i] : 0.0) + (y[i] - pv[i]) / w[i];
      for (int a = 0; a < C; ++a) {
        const double xa = X[(size_t)a * N + i] * w[i];
        b[a] += xa * z;
        for (int c = 0; c < C; ++c) A[(size_t)a * C + c] += xa * X[(size_t)c * N + i];
      }
    }
    if (!solve_dense(A, b, C, betanew)) return false;
    bool ok_search = false;
    for (int ls = 0; ls < prm.niter_max_line_search; ++ls) {
      bool inside = true;
      for (int64_t i = 0; i < N; ++i) {
        double e = offset ? offset[i] : 0.0;
        for (int a = 0; a < C; ++a) e += X[(size_t)a * N + i] * betanew[a];
        eta[i] = e;
        pv[i] = get_pvec1(e);
        if (mask[i] && !(pv[i] > 0.0 && pv[i] < 1.0)) inside = false;
      }
      dev_new = dev(pv);
      if (inside && (!check_hs_dev || dev_new < dev_old)) { ok_search = true; break; }
      for (int a = 0; a < C; ++a) betanew[a] = (beta[a] + betanew[a]) / 2;
    }
    if (!ok_search) return false;
    double smax = 0.0;
    for (int a = 0; a < C; ++a) {
      double sc = 0.0;
      for (int64_t i = 0; i < N; ++i) if (mask[i]) sc += X[(size_t)a * N + i] * (y[i] - pv[i]);
      smax = std::max(smax, std::fabs(sc));
    }
    if (smax < NUMTOL) break;
    if (!small_score && niter < 20 && smax < 1) small_score = true;
    if (small_score && niter > 20 && smax > 5) return false;
    diff_dev = std::fabs(dev_new - dev_old) / (0.1 + std::fabs(dev_new));
    beta = betanew;
    dev_old = dev_new;
  }
  if ((diff_dev == 0 || diff_dev >= NUMTOL) && niter > prm.niter_max) return false;
  if (pv_out) *pv_out = pv;
  if (beta_out) *beta_out = betanew;
  return true;
}

// Standard normal quantile (what boost::math::quantile(normal(0,1), p) returns in rint_pheno, Pheno.cpp:2002-2008):
// Wichura's algorithm AS 241 (PPND16), relative accuracy about 1e-16.
double norm_quantile(double p) {
  const double q = p - 0.5;
  if (std::fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    const double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r + 4.5921953931549871457e4) * r +
                           1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r + 1.3314166789178437745e2) * r + 3.3871328727963666080e0);
    const double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r + 2.1213794301586595867e4) * r +
                           5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r + 4.2313330701600911252e1) * r + 1.0);
    return q * num / den;
  }
  double r = q < 0 ? p : 1.0 - p;
  r = std::sqrt(-std::log(r));
  double v;
  if (r <= 5.0) {
    r -= 1.6;
    const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r + 1.27045825245236838258e0) * r +
                           3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r + 4.63033784615654529590e0) * r + 1.42343711074968357734e0);
    const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r +
                           6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r + 2.05319162663775882187e0) * r + 1.0);
    v = num / den;
  } else {
    r -= 5.0;
    const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r +
                           2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r + 5.46378491116411436990e0) * r + 6.65790464350110377720e0);
    const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r +
                           1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r + 5.99832206555887937690e-1) * r + 1.0);
    v = num / den;
  }
  return q < 0 ? -v : v;
}

// ---- time-to-event traits: the null Cox model of step 1 (fit_null_cox, Step1_Models.cpp:353-440) ------------------------------------------
// cox_ridge with lambda = 0 on the covariates (cox_ridge.cpp:8-178; survival_data::setup, survival_data.cpp:9-100): IRLS on the diagonal of
// the Hessian, one cyclic pass over the C coordinates per iteration, step halving on the deviance.  X: col-major N x C.  eta = X beta on
// the unmasked samples, 0 elsewhere.  (Level 1 -- the same model on the thousands of level-0 predictors -- runs in the library: rg_l1_cox.)
bool cox_null_fit(const double* time, const double* event, const uint8_t* mask, const double* X, int64_t N, int C, const Params& prm, std::vector<double>& eta) {
  const int64_t n = N;
  double neff = 0;
  for (int64_t i = 0; i < n; ++i) neff += mask[i];
  const double w = 1.0 / neff;
  std::vector<int64_t> ord(n);
  for (int64_t i = 0; i < n; ++i) ord[i] = i;
  auto st = [&](int64_t i) { return mask[i] ? event[i] : -999.0; };
  std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return time[a] != time[b] ? time[a] < time[b] : st(a) > st(b); });
  std::vector<uint8_t> keep(n), dd(n, 0), ev1(n, 0);
  std::vector<double> ww(n, 0.0), wsub;
  std::vector<int64_t> evs;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t s = ord[i];
    keep[i] = mask[s];
    if (mask[s] && event[s] == 1.0) { ev1[i] = dd[i] = 1; ww[i] = w; evs.push_back(i); }
  }
  for (size_t a = 0; a < evs.size();) {
    size_t b = a + 1;
    while (b < evs.size() && time[ord[evs[b]]] == time[ord[evs[a]]]) ++b;
    if (b - a > 1) { for (size_t t = a + 1; t < b; ++t) { dd[evs[t]] = 0; ww[evs[t]] = 0.0; } ww[evs[a]] = (double)(b - a) * w; }
    wsub.push_back((double)(b - a) * w);
    a = b;
  }
  double lsat = 0;
  for (double x : wsub) lsat -= x * std::log(x);
  std::vector<double> beta(C, 0.0), beta_old(C), g(n), h(n), z(n), rsk(n);
  eta.assign(n, 0.0);
  auto deviance = [&]() {
    double run = 0, ll = 0;
    for (int64_t i = n - 1; i >= 0; --i) {
      const double e = eta[ord[i]];
      if (keep[i]) run += w * std::exp(e);
      if (keep[i] && ev1[i]) ll += w * e;
      if (keep[i] && dd[i]) ll -= ww[i] * std::log(run);
    }
    return 2.0 * (lsat - ll);
  };
  auto grad = [&]() {      // coxGrad (cox_ridge.cpp:60-82): g, h in sample order
    double mean = 0;
    for (int64_t i = 0; i < n; ++i) if (mask[i]) mean += eta[i];
    mean *= w;
    double run = 0;
    for (int64_t i = n - 1; i >= 0; --i) { if (keep[i]) run += w * std::exp(eta[ord[i]] - mean); rsk[i] = run; }
    double A = 0, B = 0;
    for (int64_t i = 0; i < n; ++i) {
      if (keep[i] && dd[i]) { A += ww[i] / rsk[i]; B += ww[i] / (rsk[i] * rsk[i]); }
      const int64_t s = ord[i];
      if (keep[i]) { const double we = w * std::exp(eta[s] - mean); g[s] = w * (ev1[i] ? 1.0 : 0.0) - we * A; h[s] = we * we * B - we * A; }
      else { g[s] = 0; h[s] = 0; }
    }
  };
  auto set_eta = [&]() {
    for (int64_t i = 0; i < n; ++i) {
      double e = 0;
      if (mask[i]) for (int c = 0; c < C; ++c) e += X[(size_t)c * N + i] * beta[c];
      eta[i] = e;
    }
  };
  const double tol = 2.5e-4;        // numtol_cox, Regenie.hpp:221
  double dev_prev = deviance(), obj_prev = dev_prev;
  for (int t = 1; t <= prm.niter_max; ++t) {
    beta_old = beta;
    grad();
    for (int64_t i = 0; i < n; ++i) z[i] = (mask[i] ? eta[i] : 0.0) - (h[i] != 0 ? g[i] / h[i] : 0.0);
    for (int k = 0; k < C; ++k) {
      const double* xk = X + (size_t)k * N;
      double rx = 0, s2 = 0;
      for (int64_t i = 0; i < n; ++i) { rx += h[i] * (z[i] - eta[i]) * xk[i]; s2 += xk[i] * xk[i] * h[i]; }
      const double b1 = (rx + beta[k] * s2) / s2;                 // lambda = 0
      for (int64_t i = 0; i < n; ++i) if (mask[i]) eta[i] += xk[i] * (b1 - beta[k]);
      beta[k] = b1;
    }
    double dev = deviance(), obj = dev;
    if (dev - dev_prev > tol) {
      int ii = 0;
      while (dev - dev_prev > tol) {
        if (++ii > prm.niter_max_line_search) return false;
        for (int c = 0; c < C; ++c) beta[c] = (beta[c] + beta_old[c]) / 2;
        set_eta();
        dev = obj = deviance();
      }
    }
    double score = 0;
    for (int k = 0; k < C; ++k) {
      double sx = 0;
      for (int64_t i = 0; i < n; ++i) sx += g[i] * X[(size_t)k * N + i];
      score = std::max(score, std::fabs(sx));
    }
    const bool stop = std::fabs(obj - obj_prev) / (0.1 + std::fabs(obj)) < tol || score < tol;
    dev_prev = dev; obj_prev = obj;
    if (stop) return true;
  }
  return false;
}

// fit_null_poisson + fit_poisson (Step1_Models.cpp:225-345) for one phenotype; offset may be null (zero); eta_out = offset + X beta on
// success, pv_out (optional) the fitted rates
bool fit_poisson(const double* y, const double* X, const uint8_t* mask, int64_t N, int C, const Params& prm, std::vector<double>& eta,
                 const double* offset = nullptr, std::vector<double>* pv_out = nullptr) {
  std::vector<double> beta(C, 0.0), betanew(C, 0.0), pv(N);
  auto dev = [&](const std::vector<double>& pp) {
    double t = 0.0;
    for (int64_t i = 0; i < N; ++i) if (mask[i]) t -= y[i] * std::log(pp[i]) - pp[i];
    return 2.0 * t;
  };
  auto any_zero = [&]() { for (int64_t i = 0; i < N; ++i) if (mask[i] && pv[i] == 0.0) return true; return false; };
  eta.assign(N, 0.0);
  double esum = 0.0;
  for (int64_t i = 0; i < N; ++i) {  // starting values: p = y + 0.1, eta = log p on the analysed samples, intercept = mean(eta)
    pv[i] = y[i] + 1e-1;
    eta[i] = mask[i] ? std::log(pv[i]) : 0.0;
    esum += eta[i];
  }
  beta[0] = esum / (double)N;
  if (offset) { double osum = 0.0; for (int64_t i = 0; i < N; ++i) osum += offset[i]; beta[0] -= osum / (double)N; }   // Step1_Models.cpp:247
  double dev_old = dev(pv), dev_new = dev_old;
  int niter = 0;
  bool dev_conv = false;
  while (true) {
    if (++niter > prm.niter_max) break;
    if (any_zero()) return false;
    std::vector<double> A((size_t)C * C, 0.0), b(C, 0.0);
    for (int64_t i = 0; i < N; ++i) {
      if (!mask[i]) continue;
      const double z = eta[i] - (offset ? offset[i] : 0.0) + (y[i] - pv[i]) / pv[i];
      for (int a = 0; a < C; ++a) {
        const double xa = X[(size_t)a * N + i] * pv[i];
        b[a] += xa * z;
        for (int c = 0; c < C; ++c) A[(size_t)a * C + c] += xa * X[(size_t)c * N + i];
      }
    }
    if (!solve_dense(A, b, C, betanew)) return false;
    for (int ls = 0; ls < prm.niter_max_line_search; ++ls) {
      for (int64_t i = 0; i < N; ++i) {
        double e = offset ? offset[i] : 0.0;
        for (int a = 0; a < C; ++a) e += X[(size_t)a * N + i] * betanew[a];
        eta[i] = e;
        pv[i] = std::exp(e);
      }
      dev_new = dev(pv);
      if (!any_zero()) break;
      for (int a = 0; a < C; ++a) betanew[a] = (beta[a] + betanew[a]) / 2;
    }
    double smax = 0.0;
    for (int a = 0; a < C; ++a) {
      double sc = 0.0;
      for (int64_t i = 0; i < N; ++i) if (mask[i]) sc += X[(size_t)a * N + i] * (y[i] - pv[i]);
      smax = std::max(smax, std::fabs(sc));
    }
    dev_conv = std::fabs(dev_new - dev_old) / (0.1 + std::fabs(dev_new)) < 1e-8;  // params->tol
    if (smax < 1e-8) break;
    beta = betanew;
    dev_old = dev_new;
  }
  if (!dev_conv && niter > prm.niter_max) return false;
  if (pv_out) *pv_out = pv;
  return true;
}

void apply_sample_and_variant_filters(Run& r);
void blup_read(struct Run& r, const std::unordered_map<std::string, int64_t>& idx);

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
    if (nt < 1) nt = std::max(1, (int)std::thread::hardware_concurrency() - 1);
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
  if (!pg) {
    std::string fn = p.bed + ".fam";
    std::ifstream f(fn);
    if (!f) throw std::runtime_error("cannot open file : " + fn);
    sout << std::left << std::setw(20) << " * fam" << ": [" << fn << "] ";
    std::string line;
    std::set<std::string> seen;
    while (std::getline(f, line)) {
      auto t = split_ws(line);
      if (t.size() < 6) throw std::runtime_error("incorrectly formatted fam file at line " + std::to_string(r.fam_ids.size() + 1));
      std::string id = t[0] + "_" + t[1];
      if (!seen.insert(id).second) throw std::runtime_error("duplicate individual in fam file : FID_IID=" + id);
      if (t[4] != "0" && t[4] != "1" && t[4] != "2") throw std::runtime_error("unrecognized sex code in file : '" + t[4] + "'");
      if (t[4] == "1") r.has_male = true;
      r.fam_ids.push_back(id);
    }
    r.n_file = (int64_t)r.fam_ids.size();
    sout << "n_samples = " << r.n_file << "\n";
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
    sout << std::left << std::setw(20) << (" * " + kind) << ": [" << fn << "] ";
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
    sout << "n_snps = " << lineno << "\n";
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
      if (nt < 1) nt = std::max(1, (int)std::thread::hardware_concurrency() - 1);
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
  for (int q = 0; q < r.P; ++q) {
    if (!files.count(r.pheno_names[q])) throw std::runtime_error("No step 1 file provided for phenotype '" + r.pheno_names[q] + "'.");
    Run::Blup& bl = r.blups[q];
    bl.file = files[r.pheno_names[q]];
    sout << "   -file [" << bl.file << "] for phenotype '" << r.pheno_names[q] << "'\n";
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
    auto hdr = split_ws(line);
    if (hdr.empty() || hdr[0] != "FID_IID") throw std::runtime_error("header of blup file must start with FID_IID (=" + (hdr.empty() ? std::string() : hdr[0]) + ")");
    bl.col_sample.assign(hdr.size(), -1);
    for (size_t c = 1; c < hdr.size(); ++c) {
      auto it = idx.find(hdr[c]);
      if (it != idx.end()) bl.col_sample[c] = it->second;
    }
    bl.line_off.push_back(gzf ? 0 : (int64_t)f.tellg());
    std::getline(f, line);
    if (gzf) bl.lines.push_back(line);
    auto l2 = split_ws(line);
    if (l2.size() != hdr.size()) throw std::runtime_error("blup file for phenotype '" + r.pheno_names[q] + "' has different number of entries on line 2 compared to the header.");
    std::vector<uint8_t> have(N, 0);
    for (size_t c = 1; c < hdr.size(); ++c)
      if (bl.col_sample[c] >= 0 && convert_double(l2[c]) != MISSING) have[bl.col_sample[c]] = 1;
    int64_t before = 0, after = 0;
    for (int64_t i = 0; i < N; ++i) { before += r.mask[(size_t)q * N + i]; r.mask[(size_t)q * N + i] &= have[i]; after += r.mask[(size_t)q * N + i]; }
    if (after < 1) throw std::runtime_error("all individuals are missing LOCO predictions for phenotype '" + r.pheno_names[q] + "'.");
    if (after < before) sout << "    + " << before - after << " individuals with missing LOCO predictions will be ignored for the trait\n";
    for (;;) {   // offsets of the following lines (one per chromosome)
      const int64_t off = gzf ? 0 : (int64_t)f.tellg();
      if (!std::getline(f, line) || line.empty()) break;
      bl.line_off.push_back(off);
      if (gzf) bl.lines.push_back(line);
    }
  }
}

// a loop spread over host threads (step 2's variant loop is the reference's OpenMP loop in compute_tests_mt, Data.cpp:2484-2486)
template <class F>
void parallel_for(int n, int nthreads, F&& fn) {
  nthreads = std::max(1, std::min(nthreads, n));
  if (nthreads == 1) { for (int j = 0; j < n; ++j) fn(j); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t]() { for (int j = t; j < n; j += nthreads) fn(j); });
  for (auto& x : th) x.join();
}

void read_pheno_cov(Run& r) {  // Pheno.cpp:50-146, :148-364, :573-808, :810-841, :1903-1935
  const Params& p = r.p;
  const int64_t N = r.N;
  std::unordered_map<std::string, int64_t> idx;
  idx.reserve((size_t)N * 2);
  for (int64_t i = 0; i < N; ++i) idx[r.ids[i]] = i;
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
    std::vector<std::string> lines;
    while (std::getline(f, line)) lines.push_back(std::move(line));
    struct Rec { int64_t i; int state; };      // state 0: use, 1: blank line, 2: wrong number of columns, 3: a value that is not a number
    std::vector<Rec> recs(lines.size());
    std::vector<double> vals(lines.size() * (size_t)NV);
    {
      const int nt = std::max(1, std::min(16, (int)std::thread::hardware_concurrency() - 1));
      const int nchunk = (int)std::min<size_t>(lines.size(), (size_t)nt * 4);
      parallel_for(nchunk, nt, [&](int c) {
        for (size_t li = lines.size() * c / nchunk, le = lines.size() * (c + 1) / nchunk; li < le; ++li) {
          const auto t = split_ws(lines[li]);
          Rec& rc = recs[li];
          rc.i = -1; rc.state = 0;
          if (t.empty()) { rc.state = 1; continue; }
          if (t.size() != hdr.size()) { rc.state = 2; continue; }
          auto it = idx.find(t[0] + "_" + t[1]);
          if (it == idx.end()) continue;
          rc.i = it->second;
          try { for (int q = 0; q < NV; ++q) vals[li * (size_t)NV + q] = convert_double(t[keep_cols[q]]); }
          catch (...) { rc.state = 3; }
        }
      });
    }
    for (size_t li = 0; li < lines.size(); ++li) {
      if (recs[li].state == 1) continue;
      if (recs[li].state == 2) throw std::runtime_error("incorrectly formatted phenotype file.");
      if (recs[li].i < 0) continue;
      const int64_t i = recs[li].i;
      std::vector<std::string> t;                     // the tokens again, for the messages of the rare failing line only
      auto tok = [&]() -> const std::vector<std::string>& { if (t.empty()) t = split_ws(lines[li]); return t; };
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
        for (int64_t i = 0; i < N; ++i) ncases += (r.Yraw[(size_t)q * N + i] == 1 && r.mask[(size_t)q * N + i]);
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
    while (std::getline(f, line)) {
      auto t = split_ws(line);
      if (t.empty()) continue;
      if (t.size() != hdr.size()) throw std::runtime_error("incorrectly formatted covariate file.");
      auto it = idx.find(t[0] + "_" + t[1]);
      if (it == idx.end()) continue;
      const int64_t i = it->second;
      if (in_cov[i]) throw std::runtime_error("individual appears more than once in covariate file: FID=" + t[0] + " IID=" + t[1]);
      in_cov[i] = 1;
      for (size_t c = 0; c < kc.size(); ++c) {
        double v;
        if (is_cat[c]) {
          const std::string& tok = t[kc[c]];
          if (tok == "NA" || tok == "nan" || tok == "inf") v = MISSING;
          else {
            auto lv = levels[c].find(tok);
            if (lv == levels[c].end()) lv = levels[c].emplace(tok, (int)levels[c].size()).first;
            v = lv->second;
          }
        } else v = convert_double(t[kc[c]]);
        Xraw[(size_t)(1 + c) * N + i] = v;
        if (v == MISSING) { in_cov[i] = 0; break; }
      }
    }
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
      bool ok = fit_logistic(r.Yraw.data() + (size_t)q * N, r.X.data(), r.mask.data() + (size_t)q * N, N, nz, p, true, eta, nullptr, nullptr, &b0);
      if (!ok) ok = fit_logistic(r.Yraw.data() + (size_t)q * N, r.X.data(), r.mask.data() + (size_t)q * N, N, nz, p, false, eta, nullptr, nullptr, &b0);
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
      if (!cox_null_fit(r.Yraw.data() + (size_t)q * N, r.Yevent.data() + (size_t)q * N, r.mask.data() + (size_t)q * N, r.X.data(), N, nz, p, eta))
        throw std::runtime_error("step1 cox null regression did not converge for phenotype '" + r.pheno_names[q] + "' by coordinate descent (the reference's Newton "
                                 "fall-back, cox_firth.cpp, is not built).");
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
  sout << "done\n";
}

std::string get_fullpath(const std::string& f) {  // Data.cpp:1150-1194
  char buf[PATH_MAX];
  if (realpath(f.c_str(), buf)) return buf;
  if (!f.empty() && f[0] == '/') return f;
  if (getcwd(buf, sizeof(buf))) return std::string(buf) + "/" + f;
  return f;
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

void check(rg_ctx* ctx, int rc) {
  if (rc != 0) throw std::runtime_error(rg_last_error(ctx));
}

// -log10 p of a 1-df chi-square statistic (get_logp, Regenie.cpp:1843-1856)
double get_logp(double t) {
  if (t < 0 && std::fabs(t) < 1e-6) return 0.0;
  if (t < 0) return -1.0;
  const double pv = std::erfc(std::sqrt(t / 2.0));   // cdf(complement(chi_squared(1), t))
  const double lp = pv == 0 ? std::log10(2.0) - 0.5 * std::log10(2 * M_PI * t) - 0.5 * t * M_LOG10E : std::log10(pv);
  return -lp;
}

// ---- `--step 2`: single-variant additive tests (Data::test_snps_fast, Data.cpp:2230-2360) --------------------------------------------
// Host side: the LOCO reader (blup_read / blup_read_chr, Pheno.cpp:1241-1391, Step2_Models.cpp:51-140), per chromosome compute_res
// (Data.cpp:2386-2400) or the null logistic / Poisson (/ Firth) model of compute_res_bin / compute_res_count, the per-variant bookkeeping
// of parseSnpfromBed / parseSnpfromBGEN / readChunkFromPGENFileToG (allele counts, the MAC and INFO filters, allele frequencies, per-trait
// counts for samples with missing phenotypes) and the output lines (print_sum_stats_head / print_sum_stats_single, Step2_Models.cpp:
// 2410-2530).  Device side (include/rg_step2.h): every per-variant O(n) contraction -- the QT statistic whole (hard calls: 2-bit rows;
// dosages: uint16 rows; both on the i8 matrix cores), the sums the binary / count trait score tests are functions of.  Phenotypes may
// differ in their missing values: the library makes check_sparse_G's per-variant choice between the sparse and the dense branch of
// compute_score_qt.  The binary-trait score test and its approximate-Firth / saddlepoint corrections are library calls too (rg_s2_bt_*);
// the null models (C parameters) and the exact Firth test of flagged variants (C + 1 parameters) are fitted on the host.

// ---- approximate Firth correction of the binary-trait test (--firth --approx) ---------------------------------------------------------
// regenie reaches the maximisers below through a chain of solvers and fall-backs (fit_firth_nr, the pseudo-data IRLS of fit_firth_pseudo,
// step halving, restarts: Step2_Models.cpp:899-984, :1254-1737) that stop at |modified score| < 50 * numtol (null model) or < 2.5e-4 (per
// variant).  The penalised likelihood has one maximiser; here it is found to machine precision by Fisher scoring with step halving on
// the penalised deviance, which agrees with regenie's printed numbers to its stopping tolerance (1e-5 relative on BETA; the reference
// and its own golden file differ by as much).

// log |A| and A^-1 of a small SPD matrix (Cholesky); false when not positive definite
bool spd_logdet_inv(const std::vector<double>& A, int n, double& logdet, std::vector<double>* inv) {
  std::vector<double> L(A);
  logdet = 0.0;
  for (int j = 0; j < n; ++j) {
    double d = L[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L[(size_t)j * n + j] = d;
    logdet += 2.0 * std::log(d);
    for (int i = j + 1; i < n; ++i) {
      double v = L[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = v / d;
    }
  }
  if (inv) {
    inv->assign((size_t)n * n, 0.0);
    std::vector<double> col(n);
    for (int c = 0; c < n; ++c) {   // solve L L^T x = e_c
      for (int i = 0; i < n; ++i) { double v = i == c ? 1.0 : 0.0; for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * col[k]; col[i] = v / L[(size_t)i * n + i]; }
      for (int i = n - 1; i >= 0; --i) { double v = col[i]; for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * col[k]; col[i] = v / L[(size_t)i * n + i]; }
      for (int i = 0; i < n; ++i) (*inv)[(size_t)i * n + c] = col[i];
    }
  }
  return true;
}

// fit_firth_nr with cols_incl = nfree (Step2_Models.cpp:1267-1385): maximise l(beta) + 0.5 log |X^T W X| over the FIRST nfree coefficients (the
// others stay where they start); penalty and hat diagonal always use every column.  cols: K column pointers (sample-fastest, n each).
// beta in: start, out: the maximiser; dev_out: the penalised deviance there; inv_out (optional): (X^T W X)^-1.  false = no convergence.
bool firth_fit_cols(const double* y, const std::vector<const double*>& cols, const uint8_t* mask, const double* offset, int64_t n, int nfree, double maxstep,
                    std::vector<double>& beta, double* dev_out = nullptr, std::vector<double>* inv_out = nullptr) {
  const int K = (int)cols.size();
  std::vector<double> pv(n), w(n), A((size_t)K * K), Ainv, Afree((size_t)nfree * nfree), Afinv, score(nfree), step(K, 0.0), bnew(K), hx(K);
  auto pen_dev = [&](const std::vector<double>& b, double& dev) {
    double ll = 0.0;
    std::fill(A.begin(), A.end(), 0.0);
    for (int64_t i = 0; i < n; ++i) {
      if (!mask[i]) continue;
      double e = offset[i];
      for (int c = 0; c < K; ++c) e += cols[c][i] * b[c];
      const double pr = get_pvec1(e);
      pv[i] = pr; w[i] = pr * (1.0 - pr);
      ll -= (y[i] == 0.0) ? std::log(1.0 - pr) : std::log(pr);
      for (int a = 0; a < K; ++a) { const double xa = cols[a][i] * w[i]; for (int c = 0; c <= a; ++c) A[(size_t)a * K + c] += xa * cols[c][i]; }
    }
    for (int a = 0; a < K; ++a) for (int c = a + 1; c < K; ++c) A[(size_t)a * K + c] = A[(size_t)c * K + a];
    double logdet;
    if (!spd_logdet_inv(A, K, logdet, &Ainv)) return false;
    dev = 2.0 * ll - logdet;
    return true;
  };
  double dev;
  if (!pen_dev(beta, dev)) return false;
  for (int it = 0; it < 2000; ++it) {
    std::fill(score.begin(), score.end(), 0.0);
    for (int64_t i = 0; i < n; ++i) {
      if (!mask[i]) continue;
      double h = 0.0;                                  // h_i = w_i x_i^T (X^T W X)^-1 x_i
      for (int a = 0; a < K; ++a) { double t = 0.0; for (int c = 0; c < K; ++c) t += Ainv[(size_t)a * K + c] * cols[c][i]; hx[a] = t; }
      for (int a = 0; a < K; ++a) h += cols[a][i] * hx[a];
      h *= w[i];
      const double u = y[i] - pv[i] + h * (0.5 - pv[i]);
      for (int a = 0; a < nfree; ++a) score[a] += cols[a][i] * u;
    }
    const std::vector<double>* Finv = &Ainv;
    if (nfree < K) {                                   // the step solves with the free block of the information alone (:1311-1314)
      for (int a = 0; a < nfree; ++a) for (int c = 0; c < nfree; ++c) Afree[(size_t)a * nfree + c] = A[(size_t)a * K + c];
      double ld;
      if (!spd_logdet_inv(Afree, nfree, ld, &Afinv)) return false;
      Finv = &Afinv;
    }
    const int F = nfree < K ? nfree : K;
    double mx = 0.0;
    for (int a = 0; a < nfree; ++a) { double t = 0.0; for (int c = 0; c < nfree; ++c) t += (*Finv)[(size_t)a * F + c] * score[c]; step[a] = t; mx = std::max(mx, std::fabs(t)); }
    if (mx < 1e-10) { if (dev_out) *dev_out = dev; if (inv_out) *inv_out = Ainv; return true; }
    if (mx > maxstep) for (int a = 0; a < nfree; ++a) step[a] *= maxstep / mx;
    double dev_new = dev;
    bool ok = false;
    for (int hs = 0; hs < 60; ++hs) {
      double smx = 0.0;
      for (int a = 0; a < K; ++a) { bnew[a] = beta[a] + (a < nfree ? step[a] : 0.0); if (a < nfree) smx = std::max(smx, std::fabs(step[a])); }
      if (pen_dev(bnew, dev_new) && (dev_new < dev + 1e-12 || smx < 1e-6)) { ok = true; break; }      // (steps that small change the deviance by less than its rounding)
      for (int a = 0; a < nfree; ++a) step[a] /= 2.0;
    }
    if (!ok) return false;
    beta = bnew; dev = dev_new;
  }
  return false;
}

// fit_approx_firth_null (Step2_Models.cpp:899-984): the covariate-only penalised fit, offset = LOCO prediction.  X [C][n] sample-fastest.
bool firth_null_fit(const double* y, const double* X, const uint8_t* mask, const double* offset, int64_t n, int C, std::vector<double>& beta) {
  std::vector<const double*> cols(C);
  for (int c = 0; c < C; ++c) cols[c] = X + (size_t)c * n;
  return firth_fit_cols(y, cols, mask, offset, n, C, 25.0, beta);      // maxstep_null
}

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
  int nthreads = p.threads > 0 ? p.threads : std::max(1, (int)std::thread::hardware_concurrency() - 1);   // Regenie.cpp:1104-1106
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
    for (int q = 0; q < P; ++q) {
      Run::Blup& bl = r.blups[q];
      if (chrom < 1 || chrom > (int)bl.line_off.size()) throw std::runtime_error("blup file for phenotype '" + r.pheno_names[q] + "' has no line for chromosome " + std::to_string(chrom) + ".");
      std::string line;
      if (!bl.lines.empty()) line = bl.lines[chrom - 1];
      else {
        std::ifstream f(bl.file, std::ios::binary);
        f.seekg(bl.line_off[chrom - 1]);
        std::getline(f, line);
      }
      auto t = split_ws(line);
      if (t.size() != bl.col_sample.size())
        throw std::runtime_error("blup file for phenotype '" + r.pheno_names[q] + "' has different number of entries on line " + std::to_string(chrom + 1) + " compared to the header (=" + std::to_string(t.size()) + " vs " + std::to_string(bl.col_sample.size()) + ").");
      if (chr_str_to_int(t[0], p.nchrom) != chrom)
        throw std::runtime_error("blup file for phenotype '" + r.pheno_names[q] + "' starts with `" + t[0] + "`instead of chromosome number=" + std::to_string(chrom) + ".");
      std::vector<double> blup(N, 0.0);
      for (size_t c = 1; c < t.size(); ++c) {
        const int64_t i = bl.col_sample[c];
        if (i < 0 || !r.ain[i] || !r.mask[(size_t)q * N + i]) continue;
        const double v = convert_double(t[c]);
        if (v == MISSING) throw std::runtime_error("individual has missing predictions (chr=" + std::to_string(chrom) + ";phenotype=" + r.pheno_names[q] + ").");
        blup[i] = v;
      }
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
          ok = fit_logistic(yq, Xc.data(), mq, n, C, p, true, eta, off.data(), &pv, &bnull);
          if (!ok) ok = fit_logistic(yq, Xc.data(), mq, n, C, p, false, eta, off.data(), &pv, &bnull);
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
    if (glm) {   // compute_res_bin / compute_res_count (Data.cpp:2439-2455): the null models of the chromosome go to the device
      rg_s2_bt_null nm;
      memset(&nm, 0, sizeof(nm));
      nm.family = p.ct ? 1 : 0; nm.X = Xc.data(); nm.y = Yc.data(); nm.mask = Mc.data(); nm.fitted = bt_fit.data();
      nm.firth_offset = (firth && p.firth_approx) ? firth_off.data() : nullptr; nm.pass = bt_pass.data();
      s2check(rg_s2_bt_set_null(s2, &nm));
    } else s2check(rg_s2_set_null(s2, Xc.data(), resc.data(), Mc.data(), scf.data()));
    sout << "done (" << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - tb).count() << "ms) \n";

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
      if (in == In::PgenHard) {   // ReadHardcalls per variant (Geno.cpp:2570-2573), as .bed-coded rows (00 = two ALT copies)
        if (rg_pgen_read_bed_rows(r.pgen, bs, vidx.data(), rows.data(), r.bpr) != RG_PGEN_OK) throw std::runtime_error(rg_pgen_last_error(r.pgen));
      } else if (in == In::Dosage) {
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
      if (in == In::Dosage) {
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
          s2check(rg_s2_bt_score_int(s2, G16.data(), n, bs, 0, dscale, NUMTOL, &bo));
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
        if (integral) s2check(rg_s2_qt_block_int(s2, G16.data(), n, bs, 0, dscale, NUMTOL, &o));
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
            ln << head.str() << af << " ";
            if (show_info) ln << info << " ";
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
      sout << "done (" << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t1).count() << "ms) \n";
    }
  }
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
    for (int g = 0; g < G; ++g) if (errs[g]) std::rethrow_exception(errs[g]);
    // the parts' result files in block order -> PFX_<trait>.regenie[.gz]
    for (int q = 0; q < P; ++q) {
      const std::string fn = p.out + "_" + r.pheno_names[q] + ".regenie" + (p.gz ? ".gz" : "");
      TextOut of(fn, p.gz);
      if (!of) throw std::runtime_error("cannot write file : " + fn);
      std::vector<char> buf(8 << 20);
      for (int g = 0; g < G; ++g) {
        std::ifstream in(parts[g].files[q], std::ios::binary);
        while (in) { in.read(buf.data(), (std::streamsize)buf.size()); of.write(buf.data(), in.gcount()); }
        in.close();
        std::remove(parts[g].files[q].c_str());
      }
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

// Ring of host buffers between the .bed / .pgen reader thread and rg_l0_blocks (the block loop of Data.cpp:636-678 with the file
// read taken off the critical path).  The buffers are page-locked -- copies to the device are then asynchronous and run at the
// PCIe rate -- on a thread of their own: page-locking costs ~0.2 s per GB, so a run on one GPU starts it while the phenotype and
// covariate files are still being parsed and the reader takes each buffer as it becomes ready.
struct IngestRing {
  static constexpr int NBUF = 3;
  int per = 1;                       // SNP blocks per buffer
  bool pinned = true, failed = false;
  uint8_t* mem[NBUF] = {nullptr, nullptr, nullptr};
  std::mutex mu; std::condition_variable cv; std::deque<int> free_q;
  std::thread th;
  // total_bytes: the rows this ring will carry; blk_bytes: one block; max_per: most blocks per buffer (the library's batch size) or
  // <= 0 when not known yet; pin: 1 / 0 forces page-locked / pageable buffers, -1 page-locks only when the input is several rings long
  void start(int64_t total_bytes, int64_t blk_bytes, int max_per, int pin) {
    int64_t slot = std::max<int64_t>(16LL << 20, std::min<int64_t>(total_bytes / 8, 4LL << 30));
    if (const char* e = getenv("RG_INGEST_MB")) slot = (int64_t)std::max(1, atoi(e)) << 20;
    per = (int)std::max<int64_t>(1, slot / std::max<int64_t>(1, blk_bytes));
    if (max_per > 0) per = std::min(per, max_per);
    const int64_t bytes = (int64_t)per * blk_bytes;
    pinned = pin >= 0 ? pin != 0 : total_bytes >= 4 * NBUF * bytes;
    if (const char* e = getenv("RG_INGEST_PINNED")) pinned = atoi(e) != 0;
    th = std::thread([this, bytes]() {
      for (int i = 0; i < NBUF; ++i) {
        uint8_t* m = pinned ? (uint8_t*)rg_host_alloc(bytes) : (uint8_t*)aligned_alloc(4096, (size_t)(bytes + 4095) / 4096 * 4096);
        std::lock_guard<std::mutex> lk(mu);
        if (!m) { failed = true; cv.notify_all(); return; }
        mem[i] = m;
        free_q.push_back(i);
        cv.notify_all();
      }
    });
  }
  void release() {
    if (th.joinable()) th.join();
    for (auto& m : mem) { if (m) { if (pinned) rg_host_free(m); else free(m); } m = nullptr; }
  }
  ~IngestRing() { release(); }
};

int run(int argc, char** argv) {
  Run r;
  r.p = parse_args(argc, argv);
  const Params& p = r.p;
  sout.f.open(p.out + ".log");
  auto t_start = std::chrono::steady_clock::now();
  sout << "              |=============================|\n              |   REGENIE-AMD (step 1, HIP)  |\n              |=============================|\n\n";
  sout << "Log of output saved in file : " << p.out << ".log\n\nOptions in effect:\n";
  for (int i = 1; i < argc; ++i) sout << (argv[i][0] == '-' && argv[i][1] == '-' ? "  " : " ") << argv[i] << (i + 1 < argc && argv[i + 1][0] == '-' ? " \\\n" : "");
  sout << "\n\nFitting null model\n";
  // The HIP runtime and the device contexts come up on their own threads while this one parses the text files (bringing the
  // runtime up costs 150 - 200 ms, as much as the parsing)
  std::vector<std::future<rg_ctx*>> early_ctx;
  if (p.step == 1 && !p.split_l0)
    for (int g = 0; g < p.gpus; ++g)
      early_ctx.push_back(std::async(std::launch::async, [&p, g]() -> rg_ctx* {
        rg_ctx* c = nullptr;
        if (rg_create(&c, p.single_device ? p.device : p.device + g, nullptr) != 0) return nullptr;
        return c;
      }));
  if (p.run_l0) prep_parallel_l0(r);
  read_bim_fam(r);
  if (p.split_l0) {  // set_parallel_l0 / write_l0_master (Data.cpp:232-309): master + per-job variant lists, then exit
    std::map<int, int> cn;
    for (int c : r.snp_chrom) cn[c]++;
    std::vector<int> bsizes;     // block sizes in traversal order
    for (int c : r.chr_read) {
      const int n = cn.count(c) ? cn[c] : 0;
      const int nbc = (n + p.bsize - 1) / p.bsize;
      for (int bb = 0; bb < nbc; ++bb) bsizes.push_back(std::min(p.bsize, n - bb * p.bsize));
    }
    const int total = (int)bsizes.size();
    int njobs = p.njobs;
    sout << " * running level 0 in parallel across " << total << " genotype blocks\n";
    if (njobs <= 1) throw std::runtime_error("number of jobs must be >1.");
    if (njobs > total) { sout << "   -WARNING: Number of jobs cannot be greater than number of blocks.\n"; njobs = total; }
    sout << "   -using " << njobs << " jobs\n   -master file written to [" << p.split_file << ".master]\n";
    sout << "   -variant list files written to [" << p.split_file << "_job*.snplist]\n";
    std::ofstream mf(p.split_file + ".master");
    if (!mf) throw std::runtime_error("cannot write file : " + p.split_file + ".master");
    mf << r.snp_chrom.size() << " " << p.bsize << std::endl;
    const int nall = total / njobs, rem = total - nall * njobs;
    int b = 0; size_t scount = 0;
    for (int j = 0; j < njobs; ++j) {
      const int bt = nall + (j < rem ? 1 : 0);
      int ns = 0;
      for (int t = 0; t < bt; ++t) ns += bsizes[b++];
      const std::string fname = p.split_file + "_job" + std::to_string(j + 1);
      mf << fname << " " << bt << " " << ns << std::endl;
      std::ofstream sf(fname + ".snplist");
      if (!sf) throw std::runtime_error("cannot write file : " + fname + ".snplist");
      for (int t = 0; t < ns; ++t) sf << r.snp_ids[scount + t] << "\n";
      scount += ns;
    }
    sout << "\nEnd of run\n";
    return 0;
  }
  auto since_start = [&]() { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count(); };
  sout << "   -genotype metadata read (" << since_start() << "ms since start)\n";
  const int64_t ingest_blk_bytes = (int64_t)p.bsize * r.bpr;
  IngestRing pre_ring;
  IngestRing* pre_ring_ptr = nullptr;
  if (p.step == 1 && p.gpus == 1 && !p.run_l1 && !r.dosage_mode && r.bpr > 0) {   // one GPU: the ring is page-locked under the parsing below
    pre_ring.start((int64_t)r.snp_chrom.size() * r.bpr, ingest_blk_bytes, -1, 1);
    pre_ring_ptr = &pre_ring;
  }
  read_pheno_cov(r);
  sout << "   -phenotypes and covariates ready (" << since_start() << "ms since start)\n";
  if (p.step == 2) return run_step2_all(r, t_start);
  const int64_t N = r.N;
  const int P = r.P;

  // set_blocks (Data.cpp:311-398)
  std::map<int, int> chr_nsnp;
  for (int c : r.snp_chrom) chr_nsnp[c]++;
  struct Blk { int chrom; int64_t start; int bs; };
  std::vector<Blk> blocks;
  {
    int64_t pos = 0;
    for (int c : r.chr_read) {
      const int n = chr_nsnp.count(c) ? chr_nsnp[c] : 0;
      const int nb = (n + p.bsize - 1) / p.bsize;
      for (int bb = 0; bb < nb; ++bb) blocks.push_back({c, pos + (int64_t)bb * p.bsize, std::min(p.bsize, n - bb * p.bsize)});
      pos += n;
    }
  }
  const int B = (int)blocks.size();
  if (B == 0) throw std::runtime_error("total number of blocks must be > 0.");
  const int64_t M = p.run_l0 ? r.parallel_nGeno : (int64_t)r.snp_chrom.size();   // --run-l0: global count (Data.cpp:607)
  if (p.run_l0 && (B != r.parallel_nBlocks || (int)r.snp_chrom.size() != r.parallel_nSnps))
    throw std::runtime_error("number of blocks/variants in the job's snplist doesn't match the master file.");
  if (p.run_l1) prep_parallel_l1(r, B, (int64_t)r.snp_chrom.size());
  std::vector<double> h0 = p.setl0, h1 = p.setl1;
  auto grid = [](int n) {  // set_ridge_params (Regenie.cpp:1497-1508)
    if (n < 2) throw std::runtime_error("number of ridge parameters must be at least 2 (=" + std::to_string(n) + ")");
    std::vector<double> v(n);
    for (int i = 0; i < n; ++i) v[i] = (double)i / (n - 1);
    v[0] = 0.01; v[n - 1] = 0.99;
    return v;
  };
  if (h0.empty()) h0 = grid(p.n_ridge_l0);
  if (h1.empty()) h1 = grid(p.n_ridge_l1);
  const int R0 = (int)h0.size(), R1 = (int)h1.size();
  std::vector<double> lambda(R0);
  for (int i = 0; i < R0; ++i) lambda[i] = (double)M * (1 - h0[i]) / h0[i];  // Data.cpp:607
  sout << std::left << std::setw(20) << " * block size" << ": [" << p.bsize << "]\n";
  sout << std::left << std::setw(20) << " * # blocks" << ": [" << B << "] for " << M << " variants\n";
  sout << std::left << std::setw(20) << " * # CV folds" << ": [" << p.cv_folds << "]\n";
  if (p.loocv) sout << std::left << std::setw(20) << " * LOOCV" << ": [enabled]\n";
  sout << std::left << std::setw(20) << " * ridge data_l0" << ": [ " << R0 << " : ";
  for (double h : h0) sout << h << " ";
  sout << "]\n" << std::left << std::setw(20) << " * ridge data_l1" << ": [ " << R1 << " : ";
  for (double h : h1) sout << h << " ";
  sout << "]\n";

  bool use_loocv = p.loocv;
  if (p.bt && !use_loocv && r.n_analyzed < 5000) {  // Data.cpp:353-356
    sout << "   -WARNING: Sample size is less than 5,000 so using LOOCV instead of " << p.cv_folds << "-fold CV.\n";
    use_loocv = true;
  }
  // set_folds (Data.cpp:401-426)
  std::vector<int32_t> cv_sizes(p.cv_folds, 1);
  if (!use_loocv) {
    const int64_t target = r.n_analyzed / p.cv_folds;
    if (target < 1) throw std::runtime_error("not enough samples are present for " + std::to_string(p.cv_folds) + "-fold CV.");
    int64_t cnt = 0, cum = 0;
    int cur = 0;
    for (int64_t i = 0; i < N; ++i) {
      if (r.ain[i]) cnt++;
      if (cnt == target) { cv_sizes[cur] = (int32_t)(i - cum + 1); cum += cv_sizes[cur]; cnt = 0; cur++; }
      else if (cur == p.cv_folds - 1) { cv_sizes[cur] = (int32_t)(N - i); break; }
    }
  }

  if (!use_loocv && (p.bt || p.ct)) {  // Data.cpp:436-466: every fold needs both classes / at least one count
    int64_t start = 0;
    for (int f = 0; f < p.cv_folds; ++f) {
      for (int q = 0; q < r.P; ++q) {
        if (!r.pheno_pass[q]) continue;
        double sum = 0.0, n = 0.0;
        for (int64_t i = start; i < start + cv_sizes[f]; ++i)
          if (r.mask[(size_t)q * N + i]) { sum += r.Yraw[(size_t)q * N + i]; n += 1.0; }
        if (p.bt && (sum / n) * (1 - sum / n) < NUMTOL)
          throw std::runtime_error("one of the folds has only cases/controls for phenotype '" + r.pheno_names[q] +
                                   "'. Either use smaller #folds (option --cv) or use LOOCV (option --loocv).");
        if (p.ct && sum == 0)
          throw std::runtime_error("one of the folds has only zero counts for phenotype '" + r.pheno_names[q] +
                                   "'. Either use smaller #folds (option --cv) or use LOOCV (option --loocv).");
      }
      start += cv_sizes[f];
    }
  }

  // ---- devices: one context per GPU, one host thread per context ------------------------------------------------------
  const int G = p.gpus;
  const bool use_group = G > 1 || p.force_collectives;
  std::vector<rg_ctx*> ctxs(G, nullptr);
  rg_problem pr;
  memset(&pr, 0, sizeof(pr));
  pr.n_samples = N; pr.n_file = r.n_file; pr.n_pheno = P; pr.n_cov = r.C; pr.cv_folds = use_loocv ? 0 : p.cv_folds;
  pr.n_ridge_l0 = R0; pr.ref_first = p.ref_first; pr.n_analyzed = r.n_analyzed; pr.cv_sizes = cv_sizes.data();
  pr.lambda = lambda.data(); pr.X = r.X.data(); pr.Y = r.Y.data(); pr.mask = r.mask.data();
  pr.ind_in_analysis = r.ain.data(); pr.ind_ignore = (r.N != r.n_file) ? r.ind_ignore.data() : nullptr;
  pr.neff = r.neff.data(); pr.n_blocks_total = B; pr.max_block_size = p.bsize;
  for (int g = 0; g < G; ++g) ctxs[g] = early_ctx[g].get();
  for (int g = 0; g < G; ++g) {
    if (!ctxs[g])
      throw std::runtime_error("no MI355X / HIP device available (rg_create failed for device " + std::to_string(p.single_device ? p.device : p.device + g) + ")");
    // level-0 workspaces in proportion to what this GPU will ingest: a run over a small file asks for small batches (the
    // bytes a process allocates are set-up time, for it and for the next process on the device), a large one gets the
    // library's default of 64 GB
    const int64_t bed_bytes = (int64_t)(B / G + 1) * ingest_blk_bytes;
    check(ctxs[g], rg_set_l0_workspace(ctxs[g], 0, 0, std::max<int64_t>(6000000000LL, std::min<int64_t>(64000000000LL, 8 * bed_bytes))));
    check(ctxs[g], rg_set_problem(ctxs[g], &pr));
  }
  sout << "   -GPU context" << (G > 1 ? "s" : "") << " ready (" << since_start() << "ms since start)\n";
  rg_ctx* ctx = ctxs[0];
  rg_group* grp = nullptr;
  if (use_group) {
    if (rg_group_create(&grp, G, ctxs.data(), p.transport) != 0 || !grp) throw std::runtime_error(std::string("cannot set up the GPU group: ") + rg_last_error(ctxs[0]));
    sout << std::left << std::setw(20) << " * # GPUs" << ": [" << G << "] (" << (p.transport == RG_TRANSPORT_RCCL ? "RCCL" : "peer copies") << ")\n";
  }
  // block ranges of the ranks: floor(B/G) blocks each, the first B mod G one more (write_l0_master, Data.cpp:270-302)
  std::vector<int32_t> bbeg(G + 1, 0), pbeg(G + 1, 0);
  for (int g = 0; g < G; ++g) bbeg[g + 1] = bbeg[g] + B / G + (g < B % G ? 1 : 0);
  const bool pheno_sharded = use_group && P >= G && !p.l1_shared;      // all-to-all by phenotype; else all-gather + shared level 1
  for (int g = 0; g < G; ++g) pbeg[g + 1] = pheno_sharded ? pbeg[g] + P / G + (g < P % G ? 1 : 0) : P;
  if (!pheno_sharded) pbeg[0] = 0;

  std::mutex io_mu;      // the .pgen / .bgen readers keep per-handle state: one block read at a time
  // ---- level 0 of blocks [b_lo, b_hi) on one context -------------------------------------------------------------------
  // get_G + the block loop of level_0_calculations (Data.cpp:636-678) with the file read taken off the critical path: a
  // reader thread fills page-locked buffers (one pread per block when its variants are contiguous in the file, Geno.cpp:
  // 1702-1769 reads them one by one), the calling thread hands each buffer to rg_l0_blocks -- asynchronous copies, kernels
  // queued behind the previous batch on the other pipeline -- and recycles it once its copy has completed (rg_ingest_fence).
  auto level0_range = [&](rg_ctx* cx, int b_lo, int b_hi, std::ostringstream& lg, IngestRing* pre_ring) {
    if (b_lo >= b_hi) return;
    if (r.dosage_mode) {   // a block of dosages is bs x N_file doubles on the host: one block at a time, synchronous
      std::vector<double> dbuf;
      for (int b = b_lo; b < b_hi; ++b) {
        const Blk& bl = blocks[b];
        auto t0 = std::chrono::steady_clock::now();
        dbuf.resize((size_t)bl.bs * r.n_file);
        {
          std::lock_guard<std::mutex> lk(io_mu);
          if (r.bgenh) {  // readChunkFromBGENFileToG_fast (Geno.cpp:1574-1699): inflate + probabilities -> dosages
            if (rg_bgen_read_dosages(r.bgenh, bl.bs, &r.snp_offset[bl.start], p.ref_first ? 1 : 0, dbuf.data(), r.n_file) != RG_BGEN_OK)
              throw std::runtime_error(rg_bgen_last_error(r.bgenh));
          } else {        // Read() per kept variant (Geno.cpp:1795-1796): ALT dosages, -3 = missing
            if (rg_pgen_read_dosage_rows(r.pgen, bl.bs, &r.snp_offset[bl.start], dbuf.data(), r.n_file) != RG_PGEN_OK)
              throw std::runtime_error(rg_pgen_last_error(r.pgen));
          }
        }
        auto t1 = std::chrono::steady_clock::now();
        const int32_t id = b, bsv = bl.bs;
        const double* dp = dbuf.data();
        check(cx, rg_l0_blocks_f64(cx, 1, &id, &bsv, &dp, r.n_file, RG_MEM_HOST));
        check(cx, rg_sync(cx));
        auto t2 = std::chrono::steady_clock::now();
        lg << " block [" << b + 1 << "] (chromosome " << bl.chrom << ") : " << bl.bs << " snps  (read "
           << std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() << "ms, level 0 ridge on GPU "
           << std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() << "ms)\n";
      }
      return;
    }
    const int64_t blk_bytes = ingest_blk_bytes;
    // the ring of host buffers: the one whose page-locking was started while the text files were parsed (one GPU), or a
    // ring of this rank's own
    IngestRing own;
    IngestRing& ring = pre_ring ? *pre_ring : own;
    if (!pre_ring) own.start((int64_t)(b_hi - b_lo) * blk_bytes, blk_bytes, rg_l0_batch_blocks(cx), -1);
    const int per = ring.per;
    const int NBUF = IngestRing::NBUF;
    struct Slot { int b0 = 0, nb = 0; double read_ms = 0; };
    std::vector<Slot> slots(NBUF);
    std::mutex& mu = ring.mu; std::condition_variable& cv = ring.cv;
    std::deque<int>& free_q = ring.free_q;
    std::deque<int> ready_q;
    std::exception_ptr rd_err = nullptr;
    bool rd_done = false;
    int rd_threads = 4;    // preads of one buffer in flight (page-cache copies scale with threads; a disk queue likes depth)
    if (const char* e = getenv("RG_READ_THREADS")) rd_threads = std::max(1, atoi(e));
    std::thread reader([&]() {
      int fd = -1;
      try {
        if (!r.pgen) {
          fd = open((p.bed + ".bed").c_str(), O_RDONLY);
          if (fd < 0) throw std::runtime_error("cannot read bed file");
        }
        for (int b0 = b_lo; b0 < b_hi; b0 += per) {
          int si;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !free_q.empty() || ring.failed; });
            if (ring.failed) throw std::runtime_error("cannot allocate memory for the genotype buffers");
            si = free_q.front(); free_q.pop_front();
          }
          Slot& sl = slots[si];
          uint8_t* slmem = ring.mem[si];
          sl.b0 = b0; sl.nb = std::min(per, b_hi - b0);
          auto t0 = std::chrono::steady_clock::now();
          struct Piece { uint8_t* dst; int64_t off, want; };
          std::vector<Piece> pieces;
          for (int b = 0; b < sl.nb; ++b) {
            const Blk& bl = blocks[b0 + b];
            uint8_t* dst = slmem + (int64_t)b * blk_bytes;
            if (r.pgen) {  // ReadHardcalls per kept variant (Geno.cpp:1781-1798), as .bed-coded rows
              std::lock_guard<std::mutex> lk(io_mu);
              if (rg_pgen_read_bed_rows(r.pgen, bl.bs, &r.snp_offset[bl.start], dst, r.bpr) != RG_PGEN_OK)
                throw std::runtime_error(rg_pgen_last_error(r.pgen));
              continue;
            }
            int j = 0;
            while (j < bl.bs) {   // runs of variants that are consecutive in the file: one pread each (jumpto_bed, Geno.cpp:2828-2830)
              int e = j + 1;
              while (e < bl.bs && r.snp_offset[bl.start + e] == r.snp_offset[bl.start + e - 1] + 1) ++e;
              const int64_t want = (int64_t)(e - j) * r.bpr, off = 3 + r.snp_offset[bl.start + j] * r.bpr, chunk = 8 << 20;
              for (int64_t o = 0; o < want; o += chunk) pieces.push_back({dst + (int64_t)j * r.bpr + o, off + o, std::min(chunk, want - o)});
              j = e;
            }
          }
          if (!pieces.empty()) {
            std::atomic<int> bad{0};
            parallel_for((int)pieces.size(), rd_threads, [&](int t) {
              const Piece& pc = pieces[t];
              int64_t got = 0;
              while (got < pc.want) {
                const ssize_t k = pread(fd, pc.dst + got, (size_t)(pc.want - got), pc.off + got);
                if (k <= 0) { bad = 1; return; }
                got += k;
              }
            });
            if (bad) throw std::runtime_error("cannot read bed file");
          }
          sl.read_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          {
            std::lock_guard<std::mutex> lk(mu);
            ready_q.push_back(si);
          }
          cv.notify_all();
        }
      } catch (...) {
        std::lock_guard<std::mutex> lk(mu);
        rd_err = std::current_exception();
      }
      if (fd >= 0) close(fd);
      {
        std::lock_guard<std::mutex> lk(mu);
        rd_done = true;
      }
      cv.notify_all();
    });
    std::exception_ptr main_err = nullptr;
    try {
      for (;;) {
        int si = -1;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return !ready_q.empty() || rd_done; });
          if (!ready_q.empty()) { si = ready_q.front(); ready_q.pop_front(); }
          else if (rd_err) std::rethrow_exception(rd_err);
          else break;
        }
        Slot& sl = slots[si];
        std::vector<int32_t> ids(sl.nb), bss(sl.nb);
        std::vector<const uint8_t*> ptrs(sl.nb);
        int64_t nsnp = 0;
        for (int b = 0; b < sl.nb; ++b) {
          ids[b] = sl.b0 + b; bss[b] = blocks[sl.b0 + b].bs; ptrs[b] = ring.mem[si] + (int64_t)b * blk_bytes;
          nsnp += bss[b];
        }
        auto t1 = std::chrono::steady_clock::now();
        check(cx, rg_l0_blocks(cx, sl.nb, ids.data(), bss.data(), ptrs.data(), r.bpr, RG_MEM_HOST));
        check(cx, rg_ingest_fence(cx));     // the rows have crossed PCIe: the buffer goes back to the reader
        auto t2 = std::chrono::steady_clock::now();
        lg << " blocks [" << sl.b0 + 1 << ".." << sl.b0 + sl.nb << "] (chromosomes " << blocks[sl.b0].chrom << ".." << blocks[sl.b0 + sl.nb - 1].chrom
           << ") : " << nsnp << " snps  (read " << (int64_t)sl.read_ms << "ms in the reader thread, queued on the GPU after "
           << std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() << "ms)\n";
        {
          std::lock_guard<std::mutex> lk(mu);
          free_q.push_back(si);
        }
        cv.notify_all();
      }
    } catch (...) {
      main_err = std::current_exception();
      {   // let the reader come to its end
        std::lock_guard<std::mutex> lk(mu);
        ring.failed = true;
      }
      cv.notify_all();
    }
    reader.join();
    if (!main_err) {
      auto t1 = std::chrono::steady_clock::now();
      try { check(cx, rg_sync(cx)); } catch (...) { main_err = std::current_exception(); }
      lg << "   -level 0 ridge of blocks [" << b_lo + 1 << ".." << b_hi << "] complete (" <<
          std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t1).count() << "ms after the last batch was queued)\n";
    }
    ring.release();
    if (main_err) std::rethrow_exception(main_err);
  };

  if (p.run_l1) {
    // read_l0 (Step1_Models.cpp:1921-1987): every job file holds N x (blocks_k * R0) raw doubles, column-major,
    // block-major then ridge index; its columns go to W at block offset bstart_k
    std::vector<double> slab((size_t)N * R0);
    for (size_t k = 0; k < r.mprefix.size(); ++k)
      for (int q = 0; q < P; ++q) {
        const std::string fn = r.mprefix[k] + "_l0_Y" + std::to_string(q + 1);
        std::ifstream lf(fn, std::ios::binary | std::ios::ate);
        if (!lf) throw std::runtime_error("cannot read file : " + fn);
        const int64_t want = (int64_t)sizeof(double) * N * r.btot[k] * R0;
        if ((int64_t)lf.tellg() != want) throw std::runtime_error("file " + fn + " is not the right size.");   // Step1_Models.cpp:1962
        lf.seekg(0);
        for (int bb = 0; bb < r.btot[k]; ++bb) {
          lf.read((char*)slab.data(), sizeof(double) * slab.size());
          if (!lf) throw std::runtime_error("cannot read file : " + fn);
          check(ctx, rg_l0_set_w(ctx, r.bstart[k] + bb, q, slab.data()));
        }
      }
    sout << "   -level 0 predictors read from the job files\n";
  }
  if (p.run_l0) {  // level 0 of this job, then write_l0_file (Step1_Models.cpp:728-734): PFX_job<k>_l0_Y<ph>, and stop (Data.cpp:113-117)
    std::ostringstream lg;
    level0_range(ctx, 0, B, lg, pre_ring_ptr);
    sout << lg.str();
    std::vector<double> slab((size_t)N * R0);
    for (int q = 0; q < P; ++q) {
      const std::string fn = r.job_prefix + "_l0_Y" + std::to_string(q + 1);
      std::ofstream lf(fn, std::ios::binary);
      if (!lf) throw std::runtime_error("cannot write file : " + fn);
      for (int bb = 0; bb < B; ++bb) {
        check(ctx, rg_l0_get_w(ctx, bb, q, slab.data()));
        lf.write((const char*)slab.data(), sizeof(double) * slab.size());
      }
    }
    sout << "   -level 0 predictions written to [" << r.job_prefix << "_l0_Y*]\n";
    rg_destroy(ctx);
    sout << "\nElapsed time : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << "s\nEnd of run\n";
    return 0;
  }

  // level-1 inputs shared by the ranks
  const int L = B * R0;
  std::vector<double> tau((size_t)P * R1);
  for (int q = 0; q < P; ++q)
    for (int j = 0; j < R1; ++j)   // check_l0, Step1_Models.cpp:2115-2117
      tau[(size_t)q * R1 + j] = (double)L * (1 - h1[j]) / h1[j] * (p.bt ? 3.0 / (M_PI * M_PI) : 1.0);
  std::vector<double> ct_rate(P, 0.0);
  if (p.ct)  // Step1_Models.cpp:2101-2104: tau_j = L / log(1 + h_j / (rate (1 - h_j))); rate sums the raw column as it is
    for (int q = 0; q < P; ++q) {
      double sum = 0.0;
      for (int64_t i = 0; i < N; ++i) sum += r.Yraw[(size_t)q * N + i];
      ct_rate[q] = sum / r.neff[q];
      for (int j = 0; j < R1; ++j) tau[(size_t)q * R1 + j] = (double)L / std::log(1.0 + h1[j] / (ct_rate[q] * (1 - h1[j])));
    }
  std::vector<int32_t> cols_per_chr;
  std::vector<int> chroms;
  for (int c : r.chr_read) {
    int nb = 0;
    for (auto& bl : blocks) nb += bl.chrom == c;
    if (nb > 0) { cols_per_chr.push_back(nb * R0); chroms.push_back(c); }
  }
  const int nchr = (int)chroms.size();
  const int NCS = (p.bt || p.ct || p.t2e) ? 6 : 5;
  std::vector<int64_t> order(N);  // std::map<string,...> iteration order (Data.cpp:1934)
  for (int64_t i = 0; i < N; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return r.ids[a] < r.ids[b]; });
  std::string header = "FID_IID ";
  std::vector<int64_t> kept_order;     // the analysed samples in the writer's order
  for (int64_t i : order) if (r.ain[i]) { header += r.ids[i] + " "; kept_order.push_back(i); }
  header += "\n";
  const int fmt_threads = std::max(1, std::min(32, p.threads > 0 ? p.threads : (int)std::thread::hardware_concurrency() - 1));
  // one row of a .loco / .prs file (write_chr_row, Data.cpp:1951-1975): `<chr> v1 v2 ... \n`, NA where the phenotype is missing.
  // The values are formatted by several threads over chunks of samples; the default stream format of a double (%g, six
  // significant digits) is what std::to_chars(general, 6) produces.
  auto format_rows = [&](int nrows, const std::function<double(int, int64_t)>& value, const uint8_t* maskq, std::vector<std::string>& rows) {
    const int NCK = 16;
    const int64_t nk = (int64_t)kept_order.size();
    std::vector<std::string> piece((size_t)nrows * NCK);
    parallel_for(nrows * NCK, fmt_threads, [&](int t) {
      const int row = t / NCK, ck = t % NCK;
      std::string& o = piece[t];
      const int64_t k0 = nk * ck / NCK, k1 = nk * (ck + 1) / NCK;
      o.reserve((size_t)(k1 - k0) * 12);
      char buf[48];
      for (int64_t k = k0; k < k1; ++k) {
        const int64_t i = kept_order[k];
        if (maskq[i]) {
          const auto res = std::to_chars(buf, buf + sizeof(buf), value(row, i), std::chars_format::general, 6);
          o.append(buf, res.ptr);
          o.push_back(' ');
        } else o += "NA ";
      }
    });
    rows.assign(nrows, std::string());
    for (int row = 0; row < nrows; ++row)
      for (int ck = 0; ck < NCK; ++ck) rows[row] += piece[(size_t)row * NCK + ck];
  };

  // per-phenotype results, filled by whichever rank owns the phenotype
  std::vector<std::string> ph_log(P), ph_plist(P), ph_prslist(P), ph_firthlist(P);

  // output of one phenotype (Data::output + write_predictions, Data.cpp:956-1129, :1795-1975)
  auto emit_pheno = [&](int q, const double* cs, int bestq, int conv, const double* pq /* [nchr][N] */) {
    std::ostringstream lo;
    lo << "phenotype " << r.outnum(q) << " (" << r.pheno_names[q] << ") : \n";
    if (!conv) {  // Data.cpp:1016-1021
      lo << "Level 1 model did not converge. LOCO predictions calculations are skipped.\n\n";
      ph_log[q] = lo.str();
      return;
    }
    for (int j = 0; j < R1 && p.t2e; ++j)   // Data.cpp:1043-1049: the penalty and the held-out deviance summed over the folds
      lo << " " << std::right << std::setw(5) << tau[(size_t)q * R1 + j] << " : Deviance = " << cs[5 * R1 + j] << (j == bestq ? "<- min value" : "") << "\n";
    for (int j = 0; j < R1 && !p.t2e; ++j) {
      const double neff = r.neff[q];
      double num = cs[4 * R1 + j] - cs[0 * R1 + j] * cs[1 * R1 + j] / neff;
      const double rsq = num * num / ((cs[2 * R1 + j] - cs[0 * R1 + j] * cs[0 * R1 + j] / neff) * (cs[3 * R1 + j] - cs[1 * R1 + j] * cs[1 * R1 + j] / neff));
      const double sse = cs[2 * R1 + j] + cs[3 * R1 + j] - 2 * cs[4 * R1 + j];
      double label = (double)L / (L + (p.bt ? M_PI * M_PI / 3.0 : 1.0) * tau[(size_t)q * R1 + j]);
      if (p.ct) {  // Data.cpp:1039-1054
        const double zv = std::exp((double)L / tau[(size_t)q * R1 + j]) - 1;
        label = ct_rate[q] * zv / (1 + ct_rate[q] * zv);
      }
      lo << "  " << std::right << std::setw(5) << label << " : Rsq = " << rsq;
      if (!p.ct) lo << ", MSE = " << sse / neff;
      if (p.bt || p.ct) lo << ", -logLik/N = " << cs[5 * R1 + j] / neff;
      if (j == bestq) lo << "<- min value";
      lo << "\n";
    }
    lo << "  * making predictions...writing LOCO predictions...";
    const std::string loco_fn = p.out + "_" + std::to_string(r.outnum(q)) + ".loco" + (p.gz ? ".gz" : "");  // Data.cpp:987
    std::vector<double> tot(N, 0.0);
    for (int c = 0; c < nchr; ++c)
      for (int64_t i = 0; i < N; ++i) tot[i] += pq[(size_t)c * N + i];
    std::vector<const double*> sub(p.nchrom, nullptr);
    for (int c = 0; c < nchr; ++c) if (chroms[c] >= 1 && chroms[c] <= p.nchrom) sub[chroms[c] - 1] = pq + (size_t)c * N;
    {
      TextOut lf(loco_fn, p.gz);
      if (!lf) throw std::runtime_error("cannot write file : " + loco_fn);
      lf << header;
      std::vector<std::string> rows;
      format_rows(p.nchrom, [&](int row, int64_t i) { return tot[i] - (sub[row] ? sub[row][i] : 0.0); }, r.mask.data() + (size_t)q * N, rows);
      for (int chr = 1; chr <= p.nchrom; ++chr) lf << std::to_string(chr) << " " << rows[chr - 1] << "\n";
    }
    ph_plist[q] = r.pheno_names[q] + " " + (p.use_rel_path ? loco_fn : get_fullpath(loco_fn)) + "\n";
    if (p.print_prs) {
      const std::string prs_fn = p.out + "_" + std::to_string(r.outnum(q)) + ".prs" + (p.gz ? ".gz" : "");
      TextOut pf(prs_fn, p.gz);
      pf << header;
      std::vector<std::string> rows;
      format_rows(1, [&](int, int64_t i) { return tot[i]; }, r.mask.data() + (size_t)q * N, rows);
      pf << "0 " << rows[0] << "\n";
      ph_prslist[q] = r.pheno_names[q] + " " + (p.use_rel_path ? prs_fn : get_fullpath(prs_fn)) + "\n";
    }
    if (p.write_null_firth) {   // Data.cpp:1873-1902: the null approximate-Firth estimates of every chromosome (offset = its LOCO prediction),
                                // warm-started along the chromosomes from the null logistic estimates; read back by `--step 2 --use-null-firth`
      const std::string ffn = p.out + "_" + std::to_string(q + 1) + ".firth" + (p.gz ? ".gz" : "");
      lo << "writing null approximate Firth estimates...";
      const int Cn = r.C;
      std::vector<double> bh(r.bhat_start.begin() + (size_t)q * Cn, r.bhat_start.begin() + (size_t)(q + 1) * Cn), off(N);
      std::ostringstream body;
      bool conv = true;
      for (int chr = 1; chr <= p.nchrom && conv; ++chr) {
        for (int64_t i = 0; i < N; ++i) off[i] = tot[i] - (sub[chr - 1] ? sub[chr - 1][i] : 0.0);
        conv = firth_null_fit(r.Yraw.data() + (size_t)q * N, r.X.data(), r.mask.data() + (size_t)q * N, off.data(), N, Cn, bh);
        if (!conv) break;
        body << chr << " ";
        for (int c = 0; c < Cn; ++c) body << bh[c] << (c + 1 < Cn ? " " : "");
        body << "\n";
      }
      if (!conv) lo << "WARNING: Firth failed to converge";
      else {
        TextOut ff(ffn, p.gz);
        if (!ff) throw std::runtime_error("cannot write file : " + ffn);
        ff << body.str();
        ph_firthlist[q] = r.pheno_names[q] + " " + (p.use_rel_path ? ffn : get_fullpath(ffn)) + "\n";
      }
    }
    lo << "done\n\n";
    ph_log[q] = lo.str();
  };

  // level 1 of phenotypes [q0, q0 + nq) on one context (its view already set for a phenotype-sharded run)
  auto level1_range = [&](rg_ctx* cx, int q0, int nq, bool write_out) {
    std::vector<double> cumsum((size_t)nq * NCS * R1), pred((size_t)nq * nchr * N);
    std::vector<int32_t> best(nq), converged(nq, 1);
    const double* tq = tau.data() + (size_t)q0 * R1;
    if (p.t2e) {   // one call per trait: the library derives the penalties from the score at beta = 0 and returns them
      rg_cox_options co;
      co.niter_max = p.niter_max; co.niter_max_line_search = p.niter_max_line_search; co.niter_max_ridge = p.niter_max_ridge;
      co.niter_max_line_search_ridge = 100; co.numtol_cox = 2.5e-4; co.l1_ridge_tol = 1e-4;
      for (int q = 0; q < nq; ++q) {
        double* cq = cumsum.data() + (size_t)q * NCS * R1;
        std::fill(cq, cq + (size_t)NCS * R1, 0.0);
        check(cx, rg_l1_cox(cx, q0 + q, R1, r.Yraw.data() + (size_t)(q0 + q) * N, r.Yevent.data() + (size_t)(q0 + q) * N, r.offset.data() + (size_t)(q0 + q) * N, &co,
                            nchr, cols_per_chr.data(), tau.data() + (size_t)(q0 + q) * R1, cq + 5 * R1, &converged[q], &best[q], pred.data() + (size_t)q * nchr * N));
      }
    } else if (p.bt || p.ct) {
      rg_bt_options bo;  // Regenie.hpp:287-290 defaults; family picks ridge_logistic_level_1* or ridge_poisson_level_1*
      bo.niter_max_ridge = p.niter_max_ridge; bo.niter_max_line_search_ridge = 100; bo.niter_max_line_search = p.niter_max_line_search;
      bo.family = p.ct ? 1 : 0; bo.l1_ridge_tol = 1e-4; bo.tol = 1e-8;
      check(cx, rg_l1_bt(cx, R1, tq, r.Yraw.data() + (size_t)q0 * N, r.offset.data() + (size_t)q0 * N, &bo, nchr, cols_per_chr.data(),
                         cumsum.data(), converged.data(), best.data(), pred.data()));
      for (int q = 0; q < nq; ++q) if (!r.pheno_pass[q0 + q]) converged[q] = 0;
    } else if (use_loocv)
      check(cx, rg_l1_qt_loocv(cx, R1, tq, nchr, cols_per_chr.data(), cumsum.data(), best.data(), pred.data()));
    else
      check(cx, rg_l1_qt(cx, R1, tq, nchr, cols_per_chr.data(), cumsum.data(), best.data(), pred.data()));
    if (!write_out) return;
    for (int q = 0; q < nq; ++q)
      emit_pheno(q0 + q, cumsum.data() + (size_t)q * NCS * R1, best[q], converged[q], pred.data() + (size_t)q * nchr * N);
  };

  auto tl0 = std::chrono::steady_clock::now();
  if (!use_group) {
    if (!p.run_l1) {
      std::ostringstream lg;
      level0_range(ctx, 0, B, lg, pre_ring_ptr);
      sout << lg.str();
    }
    sout << "\n Level 1 ridge...\n";
    if (p.ct) sout << " Level 1 ridge with poisson regression...\n";
    if (p.t2e) sout << " Level 1 ridge with cox regression...\n";
    tl0 = std::chrono::steady_clock::now();
    level1_range(ctx, 0, P, true);
  } else {
    // one host thread per GPU: level 0 of the rank's blocks, the exchange, level 1 of the rank's phenotypes (phenotype-
    // sharded) or of all phenotypes with the heavy steps shared (all-gather form; level-1 models other than the K-fold
    // ridge run on rank 0 alone there)
    std::vector<std::string> rank_log(G);
    std::vector<std::exception_ptr> errs(G, nullptr);
    const bool shared_l1 = !pheno_sharded && !(p.bt || p.ct || p.t2e) && !use_loocv;
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g)
      th.emplace_back([&, g]() {
        try {
          std::ostringstream lg;
          // the exchange buffers of this rank (phenotype view, packed send buffer) are allocated before level 0 starts
          check(ctxs[g], rg_group_prepare(grp, g, bbeg.data(), pheno_sharded ? pbeg.data() : nullptr));
          level0_range(ctxs[g], bbeg[g], bbeg[g + 1], lg, nullptr);
          rank_log[g] = lg.str();
          check(ctxs[g], rg_l0_finish(grp, g, bbeg.data(), pheno_sharded ? pbeg.data() : nullptr));
          if (pheno_sharded) level1_range(ctxs[g], pbeg[g], pbeg[g + 1] - pbeg[g], true);
          else if (shared_l1 || g == 0) level1_range(ctxs[g], 0, P, g == 0);
        } catch (...) {
          // whatever failed here (reader, file, level 0, level 1): the group is broken, so that the other ranks -- waiting in
          // the exchange or in a shared level-1 all-reduce, now or later -- fail too instead of waiting for this one
          errs[g] = std::current_exception();
          rg_group_abort(grp, g);
        }
      });
    for (auto& t : th) t.join();
    for (int g = 0; g < G; ++g) {
      sout << " GPU " << g << " : blocks [" << bbeg[g] + 1 << ".." << bbeg[g + 1] << "]"
           << (pheno_sharded ? ", level 1 of phenotypes [" + std::to_string(pbeg[g] + 1) + ".." + std::to_string(pbeg[g + 1]) + "]" : std::string()) << "\n" << rank_log[g];
    }
    {  // report the rank that failed first-hand, not a peer that only noticed it
      std::exception_ptr any = nullptr;
      for (int g = 0; g < G; ++g) {
        if (!errs[g]) continue;
        if (!any) any = errs[g];
        try { std::rethrow_exception(errs[g]); }
        catch (const std::exception& e) { if (!strstr(e.what(), "another GPU")) std::rethrow_exception(errs[g]); }
        catch (...) { std::rethrow_exception(errs[g]); }
      }
      if (any) std::rethrow_exception(any);
    }
    sout << "\n Level 1 ridge...\n";
  }
  sout << "   -level 1 for " << P << " phenotype(s) done ("
       << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - tl0).count() << "ms)\n\n";

  sout << "   -predictions written (" << since_start() << "ms since start)\n";
  // output (Data.cpp:956-1129): the per-phenotype tables and file lists in phenotype order
  sout << "Output\n------\n";
  std::ofstream plist(p.out + "_pred.list"), prslist;
  if (p.print_prs) prslist.open(p.out + "_prs.list");
  for (int q = 0; q < P; ++q) {
    sout << ph_log[q];
    plist << ph_plist[q];
    if (p.print_prs) prslist << ph_prslist[q];
  }
  if (p.run_l1 && !p.keep_l0)   // rm_l0_files (Data.cpp:1131-1147)
    for (auto& pre : r.mprefix)
      for (int q = 0; q < P; ++q) std::remove((pre + "_l0_Y" + std::to_string(q + 1)).c_str());
  sout << "List of blup files written to: [" << p.out << "_pred.list]\n";
  if (p.write_null_firth) {   // Data.cpp:1102-1121
    std::ofstream fl(p.out + "_firth.list");
    for (int q = 0; q < P; ++q) fl << ph_firthlist[q];
    sout << "List of files with null Firth estimates written to: [" << p.out << "_firth.list]\n";
  }
  if (p.print_prs) sout << "List of files with whole genome PRS written to: [" << p.out << "_prs.list]\n";
  if (grp) rg_group_destroy(grp);
  for (rg_ctx* cx : ctxs) rg_destroy(cx);
  sout << "\nElapsed time : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << "s\nEnd of run\n";
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  try {
    return run(argc, argv);
  } catch (const std::exception& e) {  // Regenie.cpp:72-91
    sout << "\nERROR: " << e.what() << "\nFor more information, use option '--help' or visit the website: https://rgcgithub.github.io/regenie/\n";
    return EXIT_FAILURE;
  }
}
