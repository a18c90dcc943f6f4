// regenie-amd, the C++ host driver (see driver.h): text helpers, the option table, small utilities.
//
#include "driver.h"
#include <sched.h>
#include <cmath>
#include <cstring>

namespace rgdrv {


thread_local std::ostringstream* tl_log = nullptr;
Log sout;
bool fast_exit = false;
bool full_teardown() {
  static const bool v = []() {
    const char* td = getenv("RG_TEARDOWN");
    const bool tooling = getenv("ROCPROFILER_LIBRARY_CTOR") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROF_OUTPUT_PATH") || getenv("LD_PRELOAD") ||
                         getenv("ASAN_OPTIONS") || getenv("LLVM_PROFILE_FILE");
    return (td && atoi(td) != 0) || (tooling && !(td && atoi(td) == 0));
  }();
  return v;
}
std::mutex g_reader_mu;

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

void slurp_lines(std::istream& f, TextLines& t) {
  t.buf.clear(); t.span.clear();
  {  // a plain file knows its size: one read into the final buffer (a gzipped one does not seek: chunks)
    const std::streampos cur = f.tellg();
    std::streampos endp = -1;
    if (cur != std::streampos(-1)) { f.seekg(0, std::ios::end); endp = f.tellg(); f.clear(); f.seekg(cur); }
    if (cur != std::streampos(-1) && endp != std::streampos(-1) && endp >= cur && f) {
      t.buf.resize((size_t)(endp - cur));
      f.read(&t.buf[0], (std::streamsize)t.buf.size());
      t.buf.resize((size_t)f.gcount());
    } else {
      f.clear();
      std::vector<char> chunk(1 << 22);
      while (f) {
        f.read(chunk.data(), (std::streamsize)chunk.size());
        t.buf.append(chunk.data(), (size_t)f.gcount());
      }
    }
  }
  t.span.reserve(t.buf.size() / 64 + 16);
  const char* p0 = t.buf.data();
  const size_t n = t.buf.size();
  size_t b = 0;
  while (b < n) {
    const char* nl = (const char*)memchr(p0 + b, '\n', n - b);
    const size_t e = nl ? (size_t)(nl - p0) : n;
    t.span.emplace_back(b, e);
    b = e + 1;
  }
}

int tokenize(const char* b, const char* e, Tok* out, int maxtok) {
  int n = 0;
  while (b < e) {
    while (b < e && std::isspace((unsigned char)*b)) ++b;
    const char* j = b;
    while (j < e && !std::isspace((unsigned char)*j)) ++j;
    if (j > b) { if (n < maxtok) out[n] = Tok{b, j}; ++n; }
    b = j;
  }
  return n;
}

double convert_double_tok(const char* b, const char* e) {   // same values and the same failures as convert_double(std::string(b, e))
  const size_t len = (size_t)(e - b);
  if ((len == 2 && b[0] == 'N' && b[1] == 'A') || (len == 3 && ((b[0] == 'n' && b[1] == 'a' && b[2] == 'n') || (b[0] == 'i' && b[1] == 'n' && b[2] == 'f'))))
    return MISSING;
  {  // fast path (Clinger): [-]digits[.digits][e[+-]digits] with at most 15 significant digits and a decimal exponent within +-22 --
     // mantissa and power of ten are both exact doubles, so ONE multiplication or division gives the correctly rounded value strtod
     // gives (libstdc++'s from_chars for double goes through strtod under a locale switch: 0.3 us per number)
    const char* p = b;
    bool neg = false;
    if (p < e && *p == '-') { neg = true; ++p; }
    uint64_t m = 0;
    int nd = 0, dexp = 0;
    bool any = false, ok = true;
    while (p < e && *p >= '0' && *p <= '9') { if (nd < 19) { m = m * 10 + (uint64_t)(*p - '0'); if (m) ++nd; } else ++dexp; any = true; ++p; }
    if (p < e && *p == '.') {
      ++p;
      while (p < e && *p >= '0' && *p <= '9') { if (nd < 19) { m = m * 10 + (uint64_t)(*p - '0'); if (m) ++nd; --dexp; } any = true; ++p; }
    }
    if (any && p < e && (*p == 'e' || *p == 'E')) {
      ++p;
      bool eneg = false;
      if (p < e && (*p == '+' || *p == '-')) { eneg = *p == '-'; ++p; }
      int ex = 0, ned = 0;
      while (p < e && *p >= '0' && *p <= '9' && ned < 6) { ex = ex * 10 + (*p - '0'); ++ned; ++p; }
      if (ned == 0) ok = false;
      dexp += eneg ? -ex : ex;
    }
    if (ok && any && p == e && nd <= 15 && dexp >= -22 && dexp <= 22) {
      static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
      double d = (double)m;
      d = dexp < 0 ? d / P10[-dexp] : d * P10[dexp];
      return neg ? -d : d;
    }
  }
  return convert_double(std::string(b, e));            // anything else (more digits, a leading '+', hex floats, trailing text, ...) as before
}

IdIndex::IdIndex(const std::vector<std::string>& ids) : ids_(ids) {
  size_t cap = 16;
  while (cap < ids.size() * 2 + 2) cap <<= 1;
  slot_.assign(cap, -1);
  mask_ = cap - 1;
  for (size_t i = 0; i < ids.size(); ++i) {
    const std::string& s = ids[i];
    // the key is split at its LAST '_' for hashing only (any split gives the same bytes: the hash runs over FID, '_', IID)
    uint64_t h = hash(s.data(), s.data(), s.data(), s.data() + s.size()) & mask_;
    for (;;) {
      if (slot_[h] < 0) { slot_[h] = (int32_t)i; break; }
      if (ids[(size_t)slot_[h]] == s) break;            // a repeated id keeps its first index (as the map's operator[] overwrote: last) -- ids are unique
      h = (h + 1) & mask_;
    }
  }
}
uint64_t IdIndex::hash(const char* fb, const char* fe, const char* ib, const char* ie) {
  uint64_t h = 1469598103934665603ull;                 // FNV-1a over FID '_' IID; an empty FID part contributes nothing, no separator either
  for (const char* p = fb; p < fe; ++p) { h ^= (unsigned char)*p; h *= 1099511628211ull; }
  if (fe > fb) { h ^= (unsigned char)'_'; h *= 1099511628211ull; }
  for (const char* p = ib; p < ie; ++p) { h ^= (unsigned char)*p; h *= 1099511628211ull; }
  return h ^ (h >> 29);
}
int64_t IdIndex::find(const char* fb, const char* fe, const char* ib, const char* ie) const {
  const size_t lf = (size_t)(fe - fb), li = (size_t)(ie - ib);
  uint64_t h = hash(fb, fe, ib, ie) & mask_;
  for (;;) {
    const int32_t k = slot_[h];
    if (k < 0) return -1;
    const std::string& s = ids_[(size_t)k];
    if (s.size() == lf + 1 + li && memcmp(s.data(), fb, lf) == 0 && s[lf] == '_' && memcmp(s.data() + lf + 1, ib, li) == 0) return k;
    h = (h + 1) & mask_;
  }
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
    else if (a == "--t2e-event-l0") p.t2e_event_l0 = true;   // Regenie.cpp:366, :586
    else if (a == "--t2e-l1-pi6") p.t2e_l1_pi6 = true;       // Regenie.cpp:367, :587
    else if (a == "--covarCol" || a == "--covarColList") list(p.covar_cols, need(i));
    else if (a == "--catCovarList") list(p.cat_covar, need(i));
    else if (a == "--maxCatLevels") p.max_cat_levels = atoi(need(i).c_str());
    else if (a == "--apply-rint") p.rint = true;
    else if (a == "--keep") list(p.keep, need(i));
    else if (a == "--remove") list(p.remove, need(i));
    else if (a == "--extract") list(p.extract, need(i));
    else if (a == "--exclude") list(p.exclude, need(i));
    else if (a == "--bsize" || a == "--b") p.bsize = atoi(need(i).c_str());
    else if (a == "--nb") p.n_block = atoi(need(i).c_str());
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
  if (p.n_block < 0) usage_error("number of blocks (--nb) must be positive.");
  if (p.n_block > 0 && (p.run_l0 || p.run_l1 || p.split_l0)) {   // Regenie.cpp:1114-1119
    std::cout << "WARNING: options --split-l0/--run-l0/--run-l1 cannot be used with --nb.\n";
    p.run_l0 = p.run_l1 = p.split_l0 = false;
  }
  if (p.n_block > 0 && p.step == 2) usage_error("--nb in step 2 is not built (it limits the blocks of a step 1 run here).");
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


std::string get_fullpath(const std::string& f) {  // Data.cpp:1150-1194
  char buf[PATH_MAX];
  if (realpath(f.c_str(), buf)) return buf;
  if (!f.empty() && f[0] == '/') return f;
  if (getcwd(buf, sizeof(buf))) return std::string(buf) + "/" + f;
  return f;
}


void check(rg_ctx* ctx, int rc) {
  if (rc != 0) throw std::runtime_error(rg_last_error(ctx));
}

// Host threads worth starting: the hardware threads this process may run on (its affinity mask), and no more than twice the CPU time a
// container's cgroup grants it (cpu.max / cfs quota) -- on a 256-thread host with a 16-CPU quota, 254 workers only add scheduling and
// throttling stalls (measured on `--step 2 --bgen`: 32 workers 5.7 s, 64: 6.1 s, 254: 7.2 s).
int usable_cpus() {
  static const int n = []() {
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) hw = std::min(hw, a); }
    double quota = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
      char q[64]; long per = 0;
      if (fscanf(f, "%63s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / (double)per;
      fclose(f);
    } else {
      long q = -1, per = 0;                                                      // cgroup v1
      if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%ld", &q) != 1) q = -1; fclose(fq); }
      if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%ld", &per) != 1) per = 0; fclose(fp); }
      if (q > 0 && per > 0) quota = (double)q / (double)per;
    }
    if (quota > 0) hw = std::min(hw, std::max(2, (int)std::ceil(2.0 * quota)));
    return hw;
  }();
  return n;
}

}  // namespace rgdrv
