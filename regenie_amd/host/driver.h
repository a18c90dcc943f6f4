// regenie-amd: C++ host driver for `--step 1` / `--step 2` on MI355X: declarations shared by its translation units.
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
//   driver_common.cpp  text helpers, the option table (Regenie.cpp:146-371 subset), small utilities
//   driver_models.cpp  covariate-only null models and the host-side statistics (logistic / Poisson / Cox null fits, Firth, p-values)
//   driver_inputs.cpp  genotype metadata (.bim/.fam, .pvar/.psam, .bgen), phenotype / covariate files, LOCO files, the level-0 job files
//   driver_step2.cpp   `--step 2`: single-variant tests on the rg_step2.h kernels, one part per GPU
//   driver_step1.cpp   `--step 1`: streamed ingest, level 0, the multi-GPU exchange, level 1, the .loco / .prs writers; run()
//   driver_main.cpp    main()
#pragma once
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
#include <unordered_set>
#include <vector>
#include <charconv>
#include <climits>
#include <unistd.h>
#include <zlib.h>

#include "../../include/rg_bgen.h"
#include "../../include/rg_pgen.h"
#include "../../include/rg_step1.h"
#include "../../include/rg_step2.h"

namespace rgdrv {

constexpr double MISSING = -999.0;  // Regenie.hpp:215

struct Params {
  int step = 0;
  std::string bed, pgen, bgen, sample_file, pheno_file, covar_file, out = "regenie_out";
  std::vector<std::string> keep, remove, extract, exclude, pheno_cols, covar_cols, cat_covar;
  int max_cat_levels = 10;
  bool rint = false;
  int bsize = 0, cv_folds = 5, n_ridge_l0 = 5, n_ridge_l1 = 5, nchrom = 23, threads = 0;
  int n_block = 0;                         // --nb: total number of blocks, taken chromosome by chromosome (0: all)
  bool firth = false, firth_approx = false, firth_se = false;   // --firth --approx [--firth-se] (step 2, binary traits)
  bool write_null_firth = false;          // --write-null-firth (step 1 or 2): the null approximate-Firth estimates per chromosome, PFX_<k>.firth + PFX_firth.list
  std::string use_null_firth;             // --use-null-firth LIST (step 2): start values of the null Firth fits
  bool spa = false;                                             // --spa (step 2, binary traits)
  double pthresh = 0.05;                                        // --pThresh: score tests below it get the correction
  double min_info = 0.0; bool set_min_info = false;             // --minINFO (step 2, dosages)
  bool bt = false, ct = false, loocv = false, strict = false, ref_first = false, use_rel_path = false,
       print_prs = false, force_step1 = false, lowmem = false, force_qt = false, cc12 = false, gz = false;
  bool t2e = false;                        // --t2e: time-to-event traits (step 1: Cox ridge at level 1)
  bool t2e_event_l0 = false;               // --t2e-event-l0: which level-0 FILE a --lowmem / --run-l1 run of the reference reads for a time-to-event trait
                                           // (l0_idx, Step1_Models.cpp:2259-2261); level 1 itself fits on the time column's predictors either way (:2258,
                                           // :2269-2282) and this driver keeps them in HBM, so the switch is accepted and changes nothing -- regenie's own
                                           // in-memory outputs with and without it are byte-identical (tests/golden/ref_outputs/t2e_kfold_synth_event_l0)
  bool t2e_l1_pi6 = false;                 // --t2e-l1-pi6: level-1 penalties L (1 - h) / h * 6 / pi^2 from the heritability grid (check_l0, :2106-2110)
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
extern thread_local std::ostringstream* tl_log;
struct Log {  // mstream (Regenie.hpp:120-142): tee to stdout and <out>.log
  std::ofstream f;
  template <class T>
  Log& operator<<(const T& v) { if (tl_log) { *tl_log << v; return *this; } std::cout << v; if (f.is_open()) f << v; return *this; }
  Log& operator<<(std::ostream& (*m)(std::ostream&)) { if (tl_log) { *tl_log << m; return *this; } std::cout << m; if (f.is_open()) f << m; return *this; }
};
extern Log sout;
bool full_teardown();               // RG_TEARDOWN=1, or a tool that finalises at exit: device memory, page-locked buffers and the runtime are released in order
extern bool fast_exit;              // set at the successful end of a run: main() then leaves through _exit once the log is closed
extern std::mutex g_reader_mu;      // the .pgen / .bgen readers keep per-handle state: one block read at a time (multi-GPU step 2)

// Files (Files.cpp:38-160): a file whose name ends in ".gz" and starts with the gzip magic is read through zlib, anything
// else as plain text; `--gz` writes the .loco / .prs outputs through zlib.  (The reference needs Boost Iostreams for this.)
inline bool ends_with_gz(const std::string& fn) { return fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".gz") == 0; }
inline bool file_exists(const std::string& fn) { return access(fn.c_str(), F_OK) == 0; }

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

constexpr double NUMTOL = 1e-6;                                   // Regenie.hpp numtol
constexpr double NUMTOL_EPS = 10 * 2.220446049250313e-16;         // Regenie.hpp:225

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

// ---- driver_common.cpp
// A whole text file in memory with its line table (the remainder of a TextIn after the header line): the sample files of a 500,000-sample
// run are tokenised, matched to their samples and converted by several threads from here, without a string per token.
struct TextLines {
  std::string buf;
  std::vector<std::pair<size_t, size_t>> span;      // [begin, end) of every line, '\n' excluded (std::getline's lines)
  size_t size() const { return span.size(); }
  const char* begin(size_t l) const { return buf.data() + span[l].first; }
  const char* end(size_t l) const { return buf.data() + span[l].second; }
  std::string line(size_t l) const { return buf.substr(span[l].first, span[l].second - span[l].first); }
};
void slurp_lines(std::istream& f, TextLines& t);     // everything the stream still holds
int usable_cpus();      // hardware threads worth using: affinity mask and cgroup CPU quota taken into account
struct Tok { const char* b; const char* e; };
// whitespace-separated tokens of [b, e) (what `is >> t` would give); returns their number, stores the first `maxtok` of them
int tokenize(const char* b, const char* e, Tok* out, int maxtok);
double convert_double_tok(const char* b, const char* e);   // convert_double on a token that is not NUL-terminated
// "FID_IID" -> sample index without building the key: open addressing over the id strings
class IdIndex {
 public:
  explicit IdIndex(const std::vector<std::string>& ids);
  int64_t find(const char* fb, const char* fe, const char* ib, const char* ie) const;      // -1: no such sample
 private:
  static uint64_t hash(const char* fb, const char* fe, const char* ib, const char* ie);
  const std::vector<std::string>& ids_;
  std::vector<int32_t> slot_;
  uint64_t mask_;
};
std::vector<std::string> split_ws(const std::string& s);
std::vector<std::string> split_char(const std::string& s, char c);
int chr_str_to_int(std::string s, int nchrom);
double convert_double(const std::string& v);
std::string cpp_double(double v);
std::set<std::string> read_id_files(const std::vector<std::string>& files);
std::set<std::string> read_snp_files(const std::vector<std::string>& files);
void jacobi_eigh(std::vector<double> A, int n, std::vector<double>& d, std::vector<double>& V);
[[noreturn]] void usage_error(const std::string& m);
Params parse_args(int argc, char** argv);
std::string get_fullpath(const std::string& f);
void check(rg_ctx* ctx, int rc);
// ---- driver_models.cpp
double get_pvec1(double eta);
bool solve_dense(std::vector<double> A, std::vector<double> b, int n, std::vector<double>& x);
double norm_quantile(double p);
double get_logp(double t);
bool spd_logdet_inv(const std::vector<double>& A, int n, double& logdet, std::vector<double>* inv);
struct LogisticState { std::vector<double> beta, pv, eta; };      // what an attempt of fit_logistic leaves for the next one (the reference's in-place arguments)
bool fit_logistic(const double* y, const double* X, const uint8_t* mask, int64_t N, int C, const Params& prm, bool check_hs_dev, std::vector<double>& eta,
                  const double* offset = nullptr, std::vector<double>* pv_out = nullptr, std::vector<double>* beta_out = nullptr, LogisticState* resume = nullptr);
bool fit_poisson(const double* y, const double* X, const uint8_t* mask, int64_t N, int C, const Params& prm, std::vector<double>& eta,
                 const double* offset = nullptr, std::vector<double>* pv_out = nullptr);
bool cox_null_fit(const double* time, const double* event, const uint8_t* mask, const double* X, int64_t N, int C, const Params& prm, std::vector<double>& eta);
bool cox_null_newton(const double* time, const double* event, const uint8_t* mask, const double* X, int64_t N, int C, const Params& prm, double stephalf_tol,
                     std::vector<double>& eta);   // the reference's Newton fall-back (cox_firth.cpp without the Firth term)
bool firth_fit_cols(const double* y, const std::vector<const double*>& cols, const uint8_t* mask, const double* offset, int64_t n, int nfree, double maxstep,
                    std::vector<double>& beta, double* dev_out = nullptr, std::vector<double>* inv_out = nullptr, double stop_tol = 0.0);   // stop_tol > 0: fit_firth_nr's stopping rule
bool firth_null_fit(const double* y, const double* X, const uint8_t* mask, const double* offset, int64_t n, int C, std::vector<double>& beta);
// ---- driver_inputs.cpp
void read_bgen_meta(Run& r);
void read_bim_fam(Run& r);
void apply_sample_and_variant_filters(Run& r);
void blup_read(Run& r, const std::unordered_map<std::string, int64_t>& idx);
void read_pheno_cov(Run& r);
void prep_parallel_l0(Run& r);
void prep_parallel_l1(Run& r, int total_n_block, int64_t n_variants);
// ---- driver_step2.cpp / driver_step1.cpp
int run_step2_all(Run& r, std::chrono::steady_clock::time_point t_start);
int run(int argc, char** argv);

}  // namespace rgdrv
