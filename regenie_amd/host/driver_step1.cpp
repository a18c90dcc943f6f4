// regenie-amd, the C++ host driver (see driver.h): `--step 1` and run().
#include "driver.h"
#include "fmt_g6.h"
#include <sys/mman.h>

namespace rgdrv {

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
  // Un-pinning costs what pinning did (0.1 - 0.15 s per GB: 1.4 s for the three 4.2 GB buffers of BASELINE configs[2], measured between the
  // end of level 0 and the start of level 1 -- round 6, gpurun_out/r6_ingest): a run that leaves through _exit (the default, see the end of
  // run_step1) hands the buffers back to the system with the process instead.
  void release() {
    if (th.joinable()) th.join();
    for (auto& m : mem) { if (m && full_teardown()) { if (pinned) rg_host_free(m); else free(m); } m = nullptr; }
  }
  ~IngestRing() { release(); }
};

// The .bed itself as the source of the host -> device copies: the file is mapped and the mapping registered (read-only) with the runtime on a
// thread of its own while the text files are parsed; the pages of the page cache are then what the DMA engines read -- no pread, no host
// buffer, no page-locking of 4 GB slots (which, at 5 GB/s, also slowed the parsing it ran beside).  tools/ingest_probe.cpp: registering an
// 8 GB mapping takes 0.08 s and copies from it run at 57 GB/s.  Used for one GPU when every block's variants are consecutive in the file
// (no --extract / --exclude gaps inside a block); RG_INGEST_MAP=0, a mapping or a registration that fails, fall back on the ring.
struct BedMap {
  int fd = -1;
  uint8_t* base = nullptr;
  size_t bytes = 0;
  int state = 0;                       // 0 not started / unusable, 1 registering, 2 registered
  std::thread th;
  void start(const std::string& path, bool want_registered) {
    fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return;
    const off_t sz = lseek(fd, 0, SEEK_END);
    if (sz <= 3) return;
    void* m = mmap(nullptr, (size_t)sz, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) return;
    base = (uint8_t*)m; bytes = (size_t)sz; state = 1;
    // RG_INGEST_MAP=2: the mapping is NOT registered -- the copies then go through the runtime's own staging of pageable memory
    registered = getenv("RG_INGEST_MAP") ? atoi(getenv("RG_INGEST_MAP")) != 2 : want_registered;
    if (!registered) { state = 2; return; }
    th = std::thread([this]() { if (rg_host_register(base, (int64_t)bytes, 1) != 0) state = -1; else state = 2; });
  }
  bool registered = true;
  bool ready() { if (th.joinable()) th.join(); return state == 2; }
  ~BedMap() {
    if (th.joinable()) th.join();
    if (fast_exit) return;                 // the process is about to leave through _exit: the mapping goes with it
    if (state == 2 && registered) rg_host_unregister(base);
    if (base) munmap(base, bytes);
    if (fd >= 0) close(fd);
  }
};

// The whole .bed staged in HBM (rg_stage_*, include/rg_step1.h): from the moment the runtime is up -- under the parsing of the phenotype and
// covariate files and under rg_set_problem -- the file is copied to the device, and level 0 reads the rows in place.  A few threads pread
// 32 MB pieces of the file into a small ring of page-locked slots (8 x 32 MB: page-locking them costs 50 ms, not the 1.4 s of the 12.6 GB
// ring of round 5), one thread sends the pieces in file order (DMA from page-locked memory: the PCIe rate, where the runtime's staging of
// pageable memory gave 36 - 38 GB/s).  At BASELINE configs[2] (62.5 GB) the copy starts 0.3 s earlier than the first batch of the
// streamed form did and the level-0 thread queues kernels only.  One GPU, a file of at least 4 GB and at most a quarter of the free device memory whose kept
// variants are one range covering at least 80 % of it; RG_INGEST_STAGE=0 streams the file batch by batch as before, =2 copies from the
// mapping instead of the slots.  A copy that fails leaves the bytes behind `done` unusable: level 0 takes the rest from the mapping.
struct BedStage {
  static constexpr int NS = 8;
  rg_ctx* ctx = nullptr;
  uint8_t* dev = nullptr;
  int64_t bytes = 0, piece = 32LL << 20, npieces = 0;
  std::atomic<int64_t> done{0}, next_piece{0};
  std::atomic<int> state{0};           // 0 not used, 1 copying, 2 complete, -1 failed / refused
  std::atomic<bool> stop{false};
  std::mutex mu;
  std::condition_variable cv;
  std::thread th;
  std::vector<std::thread> fillers;
  uint8_t* slot[NS] = {};
  int64_t filled[NS] = {};             // (mu) piece number + 1 held by the slot, 0 = free
  int64_t sent = 0;                    // (mu) pieces in device memory
  int fd = -1;
  int nfill = 0;
  std::chrono::steady_clock::time_point t_first, t_last;
  void fail() { { std::lock_guard<std::mutex> lk(mu); state = -1; } cv.notify_all(); }
  void fill_loop() {
    for (;;) {
      const int64_t k = next_piece++;
      if (k >= npieces) return;
      const int si = (int)(k % NS);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return sent + NS > k || state < 0 || stop; });      // piece k - NS has left the slot
        if (state < 0 || stop) return;
      }
      if (!slot[si] && !(slot[si] = (uint8_t*)rg_host_alloc(piece))) return fail();
      const int64_t off = k * piece, n = std::min(piece, bytes - off);
      for (int64_t got = 0; got < n;) {
        const ssize_t rd = pread(fd, slot[si] + got, (size_t)(n - got), (off_t)(off + got));
        if (rd <= 0) return fail();
        got += rd;
      }
      { std::lock_guard<std::mutex> lk(mu); filled[si] = k + 1; }
      cv.notify_all();
    }
  }
  void start(std::shared_future<rg_ctx*> fctx, const std::string& path, const uint8_t* mapped, int64_t nbytes, int mode, int threads) {
    bytes = nbytes;
    npieces = (bytes + piece - 1) / piece;
    nfill = threads;
    state = 1;
    if (mode != 2) fd = open(path.c_str(), O_RDONLY);
    const bool from_map = fd < 0;
    th = std::thread([this, fctx, mapped, from_map]() {
      ctx = fctx.get();
      if (!ctx) return fail();
      dev = (uint8_t*)rg_stage_alloc(ctx, bytes, 0.25);
      if (!dev) return fail();
      t_first = std::chrono::steady_clock::now();
      if (!from_map)
        for (int t = 0; t < nfill; ++t) fillers.emplace_back([this]() { fill_loop(); });
      for (int64_t k = 0; k < npieces; ++k) {
        const int64_t off = k * piece, n = std::min(piece, bytes - off);
        const int si = (int)(k % NS);
        const uint8_t* src = mapped + off;
        if (!from_map) {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return filled[si] == k + 1 || state < 0 || stop; });
          if (filled[si] != k + 1) { lk.unlock(); return fail(); }
          src = slot[si];
        }
        if (stop || rg_stage_copy(ctx, dev + off, src, n) != 0) return fail();
        { std::lock_guard<std::mutex> lk(mu); filled[si] = 0; sent = k + 1; done = off + n; }
        cv.notify_all();
      }
      t_last = std::chrono::steady_clock::now();
      { std::lock_guard<std::mutex> lk(mu); state = 2; }
      cv.notify_all();
    });
  }
  // the stage is given up (W, the workspaces and level 1 would not fit beside it): the copy stops, the buffer goes back, level 0 streams the file
  void cancel() {
    if (state == 0) return;
    stop = true;
    cv.notify_all();
    if (th.joinable()) th.join();
    for (auto& f : fillers) if (f.joinable()) f.join();
    fillers.clear();
    { std::lock_guard<std::mutex> lk(mu); state = -1; done = 0; }      // (nothing of the buffer counts as arrived any more)
    if (dev) { rg_stage_free(ctx, dev); dev = nullptr; }
  }
  // true once bytes [0, upto) are in device memory; false when the copy failed (or was never started)
  bool wait(int64_t upto) {
    if (state == 0) return false;
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done >= upto || state < 0; });
    return done >= upto;
  }
  ~BedStage() {
    stop = true;                        // (a run that failed before level 0 was through does not wait for the rest of the file)
    cv.notify_all();
    if (th.joinable()) th.join();
    for (auto& f : fillers) if (f.joinable()) f.join();
    if (fd >= 0) close(fd);
    if (full_teardown()) {
      for (auto& m : slot) if (m) rg_host_free(m);
      if (dev) rg_stage_free(ctx, dev);
    }
  }
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
  std::vector<std::shared_future<rg_ctx*>> early_ctx;
  if (p.step == 1 && !p.split_l0)
    for (int g = 0; g < p.gpus; ++g)
      early_ctx.push_back(std::async(std::launch::async, [&p, g]() -> rg_ctx* {
        rg_ctx* c = nullptr;
        if (rg_create(&c, p.single_device ? p.device : p.device + g, nullptr) != 0) return nullptr;
        return c;
      }).share());
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
  BedMap bed_map;
  BedStage bed_stage;                  // declared after bed_map: its thread reads the mapping and is joined first
  if (p.step == 1 && !p.run_l1 && !r.dosage_mode && r.bpr > 0) {   // the host side of the ingest is set up under the parsing below
    // The mapped .bed is the source of the copies (rounds 3 - 5: registered with the runtime, and only for files up to 16 GB -- for the 62.5 GB
    // file of configs[2] the registered mapping and the ring of page-locked buffers traded places from box to box: DESIGN_HISTORY.md).
    // Round 6: ONE GPU takes the mapping UNREGISTERED, whatever the file's size -- the copies then go through the runtime's staging of pageable
    // memory, and nothing is page-locked: at configs[2] the page-locking of the ring (12.6 GB, on a thread beside the parsing) delayed the
    // runtime's start-up by 1.4 s and cost 1.4 s again when it was released; measured on a device whose memory was clean: ring 5.6 - 6.1 s,
    // registered mapping 4.4 s, unregistered mapping 3.55 s (level 0 of the 62.5 GB file queued in 1.63 s = 38 GB/s, context ready after
    // 0.42 s); configs[1]: predictions written after 0.17 - 0.19 s instead of 0.26 - 0.31 s (gpurun_out/r6_ingest, tools/r6_ingest_small.sh).
    // RG_INGEST_MAP=1 registers the mapping (several GPUs: the ranks share it and their copies are asynchronous), =0 takes the ring.
    const char* em = getenv("RG_INGEST_MAP");
    const bool want_map = em ? atoi(em) != 0 : true;
    if (!r.pgen && want_map) bed_map.start(p.bed + ".bed", p.gpus > 1);           // the mapped file, registered on its own thread (shared by the ranks)
    {
      const char* es = getenv("RG_INGEST_STAGE");
      const int64_t kept_bytes = (int64_t)r.snp_chrom.size() * r.bpr;
      if (p.gpus == 1 && !p.force_collectives && bed_map.state == 2 && !bed_map.registered && !(es && atoi(es) == 0) && !early_ctx.empty() &&
          (int64_t)bed_map.bytes >= ((es && atoi(es) != 0) ? 4 : (4LL << 30)) && 5 * kept_bytes >= 4 * ((int64_t)bed_map.bytes - 3) && !r.snp_offset.empty() &&      // (a file of a few GB is in HBM before the ring's slots are page-locked: configs[1], 1.25 GB, 0.20 s without the stage, 0.24 s with it; RG_INGEST_STAGE=1 forces it)
          r.snp_offset.back() - r.snp_offset.front() + 1 == (int64_t)r.snp_offset.size())      // the kept variants: one range of the file
        bed_stage.start(early_ctx[0], p.bed + ".bed", bed_map.base, (int64_t)bed_map.bytes, es ? atoi(es) : 1,
                        getenv("RG_STAGE_THREADS") ? std::max(1, atoi(getenv("RG_STAGE_THREADS"))) : std::max(2, std::min(6, usable_cpus() / 3)));     // 2 threads fed 24 GB/s, 4 and 8 the 50 GB/s the DMA takes (gpurun_out/r6_ingest)
    }
    if (bed_map.state == 0 && p.gpus == 1) {                                         // else, one GPU: the ring of page-locked buffers
      pre_ring.start((int64_t)r.snp_chrom.size() * r.bpr, ingest_blk_bytes, -1, 1);
      pre_ring_ptr = &pre_ring;
    }
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
    int left = p.n_block;      // --nb: at most that many blocks, taken chromosome by chromosome; the variants past them are not analysed
    for (int c : r.chr_read) {
      const int n = chr_nsnp.count(c) ? chr_nsnp[c] : 0;
      int nb = (n + p.bsize - 1) / p.bsize;
      if (p.n_block > 0) { nb = std::min(nb, left); left -= nb; }
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
  // --setl0 / --setl1 (get_unit_params, Regenie.cpp:1477-1495): sorted, duplicates removed, every value inside (0, 1)
  auto unit_params = [](std::vector<double> v, const char* opt) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (double x : v) if (!(x > 0.0 && x < 1.0)) throw std::runtime_error(std::string("must specify values for ") + opt + " in (0,1).");
    return v;
  };
  std::vector<double> h0 = unit_params(p.setl0, "--l0"), h1 = unit_params(p.setl1, "--l1");
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
    if (g == 0 && bed_stage.state != 0) {
      // W ([B R0][P][Np] doubles), the level-0 workspaces and an allowance for level 1 must still fit beside the staged file -- known only now
      // that the phenotypes are parsed; if they do not, the stage is given up and the file streams batch by batch
      const double need = (double)B * R0 * P * (double)(N + 4096) * 8.0 + (double)std::max<int64_t>(6000000000LL, std::min<int64_t>(64000000000LL, 8 * bed_bytes)) + 24e9;
      if (!rg_stage_fits(ctxs[g], (int64_t)need) || getenv("RG_STAGE_FORCE_CANCEL")) {      // (the switch: for the test of this path)
        bed_stage.cancel();
        sout << "   -the genotype file is not kept in device memory (" << (int64_t)(need / 1e9) << " GB of predictors and workspaces take the room)\n";
      }
    }
    const auto tsp = std::chrono::steady_clock::now();
    check(ctxs[g], rg_set_problem(ctxs[g], &pr));
    if (getenv("RG_TIMING"))
      fprintf(stderr, "[timing] rg_set_problem (GPU %d): %.0f ms (entered %lld ms since start)\n", g,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tsp).count(),
              (long long)std::chrono::duration_cast<std::chrono::milliseconds>(tsp - t_start).count());
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
  const bool mapped_ok = bed_map.state != 0 && bed_map.ready();     // joined here, once: the rank threads only read the flag
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
    if (!pre_ring && mapped_ok) {   // the mapped .bed: every block is a range of the file
      bool consecutive = true;
      for (int b = b_lo; b < b_hi && consecutive; ++b)
        for (int j = 1; j < blocks[b].bs; ++j)
          if (r.snp_offset[blocks[b].start + j] != r.snp_offset[blocks[b].start + j - 1] + 1) { consecutive = false; break; }
      const int64_t last = r.snp_offset[blocks[b_hi - 1].start + blocks[b_hi - 1].bs - 1];
      if (consecutive && 3 + (last + 1) * r.bpr <= (int64_t)bed_map.bytes) {
        const int nb = b_hi - b_lo;
        std::vector<int32_t> ids(nb), bss(nb);
        std::vector<const uint8_t*> ptrs(nb);
        int64_t nsnp = 0;
        for (int b = 0; b < nb; ++b) {
          ids[b] = b_lo + b; bss[b] = blocks[b_lo + b].bs; nsnp += bss[b];
          ptrs[b] = bed_map.base + 3 + r.snp_offset[blocks[b_lo + b].start] * r.bpr;
        }
        auto t1 = std::chrono::steady_clock::now();
        int b_staged = 0;          // blocks taken from the copy of the file in device memory: one library batch at a time, as the copy reaches them
        if (bed_stage.state != 0) {
          const int per = std::max(1, (int)rg_l0_batch_blocks(cx));
          std::vector<const uint8_t*> dptrs(nb);
          for (int b0 = 0; b0 < nb; b0 += per) {
            const int n = std::min(per, nb - b0);
            const int64_t end = (int64_t)(ptrs[b0 + n - 1] - bed_map.base) + (int64_t)bss[b0 + n - 1] * r.bpr;
            if (!bed_stage.wait(end)) break;
            for (int b = b0; b < b0 + n; ++b) dptrs[b] = bed_stage.dev + (ptrs[b] - bed_map.base);
            check(cx, rg_l0_blocks(cx, n, ids.data() + b0, bss.data() + b0, dptrs.data() + b0, r.bpr, RG_MEM_DEVICE));
            b_staged = b0 + n;
          }
        }
        if (b_staged < nb)
          check(cx, rg_l0_blocks(cx, nb - b_staged, ids.data() + b_staged, bss.data() + b_staged, ptrs.data() + b_staged, r.bpr, RG_MEM_HOST));
        auto t2 = std::chrono::steady_clock::now();
        lg << " blocks [" << b_lo + 1 << ".." << b_hi << "] (chromosomes " << blocks[b_lo].chrom << ".." << blocks[b_hi - 1].chrom << ") : " << nsnp
           << " snps  (" << (b_staged == nb ? "read from the copy of the file in device memory" : b_staged ? "partly read from the copy of the file in device memory" : "copied to the GPU from the mapped file")
           << ", queued after " << std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() << "ms)\n";
        if (b_staged && bed_stage.state == 2 && getenv("RG_TIMING"))
          fprintf(stderr, "[timing] the .bed in device memory: first piece requested %.0f ms, last piece arrived %.0f ms since start (%.1f GB/s)\n",
                  std::chrono::duration<double, std::milli>(bed_stage.t_first - t_start).count(), std::chrono::duration<double, std::milli>(bed_stage.t_last - t_start).count(),
                  bed_stage.bytes / 1e9 / std::max(1e-9, std::chrono::duration<double>(bed_stage.t_last - bed_stage.t_first).count()));
        check(cx, rg_sync(cx));
        lg << "   -level 0 ridge of blocks [" << b_lo + 1 << ".." << b_hi << "] complete (" <<
            std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t2).count() << "ms after the last batch was queued)\n";
        return;
      }
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
    // preads of one buffer in flight (page-cache copies scale with threads; a disk queue likes depth): half the CPUs this process may use
    // (affinity mask and cgroup quota), between 4 and 16, shared by the ranks of a multi-GPU run
    int rd_threads = std::max(4, std::min(16, usable_cpus() / 2 / std::max(1, p.gpus)));
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
      tau[(size_t)q * R1 + j] = (double)L * (1 - h1[j]) / h1[j] * (p.bt ? 3.0 / (M_PI * M_PI) : p.t2e && p.t2e_l1_pi6 ? 6.0 / (M_PI * M_PI) : 1.0);
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
  const int fmt_threads = std::max(1, std::min(32, p.threads > 0 ? p.threads : usable_cpus() - 1));
  const bool timing_on = getenv("RG_TIMING") != nullptr;
  // one row of a .loco / .prs file (write_chr_row, Data.cpp:1951-1975): `<chr> v1 v2 ... \n`, NA where the phenotype is missing.
  // The values are formatted by several threads over chunks of samples, in the default stream format of a double (%g, six
  // significant digits: rgfmt::fmt_g6, a third of the time of std::to_chars(general, 6) and the same text); the pieces go to the
  // file in order, without being joined first (115 MB per .loco file at 500,000 samples).
  const int NCK = 16;
  auto format_rows = [&](int nrows, auto&& value, const uint8_t* maskq, std::vector<std::string>& piece) {
    const int64_t nk = (int64_t)kept_order.size();
    piece.assign((size_t)nrows * NCK, std::string());
    parallel_for(nrows * NCK, fmt_threads, [&](int t) {
      const int row = t / NCK, ck = t % NCK;
      const int64_t k0 = nk * ck / NCK, k1 = nk * (ck + 1) / NCK;
      std::string& o = piece[t];
      o.resize((size_t)(k1 - k0) * 14 + 32);          // a value is at most 13 characters ("-1.23456e-308") and a blank
      char* w = &o[0];
      for (int64_t k = k0; k < k1; ++k) {
        const int64_t i = kept_order[k];
        if (maskq[i]) w = rgfmt::fmt_g6(value(row, i), w);
        else { w[0] = 'N'; w[1] = 'A'; w += 2; }
        *w++ = ' ';
      }
      o.resize((size_t)(w - &o[0]));
    });
  };
  auto write_row = [&](std::ostream& f, const std::string& label, const std::vector<std::string>& piece, int row) {
    f << label << " ";
    for (int ck = 0; ck < NCK; ++ck) f.write(piece[(size_t)row * NCK + ck].data(), (std::streamsize)piece[(size_t)row * NCK + ck].size());
    f << "\n";
  };

  // per-phenotype results, filled by whichever rank owns the phenotype
  std::vector<std::string> ph_log(P), ph_plist(P), ph_prslist(P), ph_firthlist(P);

  // output of one phenotype (Data::output + write_predictions, Data.cpp:956-1129, :1795-1975)
  auto emit_pheno = [&](int q, const double* cs, int bestq, int conv, const double* pq /* [nchr][N] */) {
    if (!r.pheno_pass[q]) return;   // null model did not converge: the trait is ignored, no table, no file, no list entry (Data.cpp:984)
    std::ostringstream lo;
    lo << "phenotype " << r.outnum(q) << " (" << r.pheno_names[q] << ") : ";
    if (!conv) {  // Data.cpp:1009-1014: on the header's own line
      lo << "Level 1 model did not converge. LOCO predictions calculations are skipped.\n\n";
      ph_log[q] = lo.str();
      return;
    }
    lo << "\n";
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
      const auto te0 = std::chrono::steady_clock::now();
      TextOut lf(loco_fn, p.gz);
      if (!lf) throw std::runtime_error("cannot write file : " + loco_fn);
      lf << header;
      std::vector<std::string> rows;
      format_rows(p.nchrom, [&](int row, int64_t i) { return tot[i] - (sub[row] ? sub[row][i] : 0.0); }, r.mask.data() + (size_t)q * N, rows);
      const auto te1 = std::chrono::steady_clock::now();
      for (int chr = 1; chr <= p.nchrom; ++chr) write_row(lf, std::to_string(chr), rows, chr - 1);
      if (timing_on) {
        const auto te2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing] .loco of phenotype %d: totals + header + formatting %.0f ms, writing %.0f ms (%d formatting threads)\n", q + 1,
                std::chrono::duration<double, std::milli>(te1 - te0).count(), std::chrono::duration<double, std::milli>(te2 - te1).count(), fmt_threads);
      }
    }
    ph_plist[q] = r.pheno_names[q] + " " + (p.use_rel_path ? loco_fn : get_fullpath(loco_fn)) + "\n";
    if (p.print_prs) {
      const std::string prs_fn = p.out + "_" + std::to_string(r.outnum(q)) + ".prs" + (p.gz ? ".gz" : "");
      TextOut pf(prs_fn, p.gz);
      pf << header;
      std::vector<std::string> rows;
      format_rows(1, [&](int, int64_t i) { return tot[i]; }, r.mask.data() + (size_t)q * N, rows);
      write_row(pf, "0", rows, 0);
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
  // One context, no exchanged view (the usual single-GPU run): the phenotypes are taken ONE AT A TIME (rg_set_l1_view with a sub-range on
  // the context's own W) and each one's tables and files are written on a thread of their own while the next phenotype is on the GPU --
  // at 500,000 samples a .loco file is 115 MB of text, formatting and writing ten of them took as long as level 1 itself.
  // The predictions of a phenotype ([nchr][N] doubles, 92 MB at BASELINE configs[2]) land in one of three page-locked slots -- the GPU fills one
  // while the writers of the two phenotypes before it read theirs.  The slots are allocated on a thread of their own while level 0 streams
  // (l1_slots_start below): page-locking all ten phenotypes' 920 MB at the head of level 1 cost 0.25 s of its 0.96 s (RG_TIMING=1).
  const int L1_SLOTS = 3;
  std::future<double*> l1_slots;
  auto l1_slots_start = [&]() {
    const int64_t bytes = (int64_t)sizeof(double) * nchr * N * std::min(P, L1_SLOTS);
    l1_slots = std::async(std::launch::async, [bytes]() { return (double*)rg_host_alloc(bytes); });
  };
  auto level1_pipelined = [&](rg_ctx* cx) {
    const size_t per = (size_t)nchr * N;
    const int nslot = std::min(P, L1_SLOTS);
    const auto ta0 = std::chrono::steady_clock::now();
    if (!l1_slots.valid()) l1_slots_start();
    double* pred = l1_slots.get();             // page-locked if the runtime grants it (device -> host at the PCIe rate, no first-touch faults),
    std::unique_ptr<double[]> pred_own;        // uninitialised either way -- every entry is written by the library
    const bool pinned = pred != nullptr;
    if (!pinned) { pred_own.reset(new double[per * nslot]); pred = pred_own.get(); }
    if (timing_on)
      fprintf(stderr, "[timing] level 1: waited %.0f ms for the %d prediction slots (%s)\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta0).count(), nslot, pinned ? "page-locked" : "pageable");
    std::vector<double> cumsum((size_t)P * NCS * R1, 0.0);
    std::vector<int32_t> best(P, 0), converged(P, 1);
    std::vector<std::future<void>> writers;
    std::exception_ptr err = nullptr;
    try {
      for (int q = 0; q < P; ++q) {
        double* cq = cumsum.data() + (size_t)q * NCS * R1;
        double* pq = pred + (size_t)(q % nslot) * per;
        const auto tw0 = std::chrono::steady_clock::now();
        if (q >= nslot) writers[q - nslot].wait();      // the slot's previous phenotype is on disk (at most nslot - 1 writers in flight behind the GPU)
        const auto tq0 = std::chrono::steady_clock::now();
        if (p.t2e) {
          rg_cox_options co{};
          co.niter_max = p.niter_max; co.niter_max_line_search = p.niter_max_line_search; co.niter_max_ridge = p.niter_max_ridge;
          co.niter_max_line_search_ridge = 100; co.numtol_cox = 2.5e-4; co.l1_ridge_tol = 1e-4;
          co.tau = p.t2e_l1_pi6 ? tau.data() + (size_t)q * R1 : nullptr;
          if (!r.pheno_pass[q]) { converged[q] = 0; best[q] = 0; }
          else check(cx, rg_l1_cox(cx, q, R1, r.Yraw.data() + (size_t)q * N, r.Yevent.data() + (size_t)q * N, r.offset.data() + (size_t)q * N, &co,
                                   nchr, cols_per_chr.data(), tau.data() + (size_t)q * R1, cq + 5 * R1, &converged[q], &best[q], pq));
        } else {
          check(cx, rg_set_l1_view(cx, nullptr, q, 1));
          const double* tq = tau.data() + (size_t)q * R1;
          if (p.bt || p.ct) {
            rg_bt_options bo{};
            bo.niter_max_ridge = p.niter_max_ridge; bo.niter_max_line_search_ridge = 100; bo.niter_max_line_search = p.niter_max_line_search;
            bo.family = p.ct ? 1 : 0; bo.l1_ridge_tol = 1e-4; bo.tol = 1e-8;
            check(cx, rg_l1_bt(cx, R1, tq, r.Yraw.data() + (size_t)q * N, r.offset.data() + (size_t)q * N, &bo, nchr, cols_per_chr.data(),
                               cq, &converged[q], &best[q], pq));
            if (!r.pheno_pass[q]) converged[q] = 0;
          } else if (use_loocv)
            check(cx, rg_l1_qt_loocv(cx, R1, tq, nchr, cols_per_chr.data(), cq, &best[q], pq));
          else
            check(cx, rg_l1_qt(cx, R1, tq, nchr, cols_per_chr.data(), cq, &best[q], pq));
        }
        const int bq = best[q], cv = converged[q];
        if (timing_on)
          fprintf(stderr, "[timing] level 1 of phenotype %d: waited %.0f ms for its slot, on the GPU %.0f ms\n", q + 1,
                  std::chrono::duration<double, std::milli>(tq0 - tw0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count());
        writers.push_back(std::async(std::launch::async, [&, q, cq, bq, cv, pq]() { emit_pheno(q, cq, bq, cv, pq); }));
      }
    } catch (...) { err = std::current_exception(); }
    if (!p.t2e) rg_set_l1_view(cx, nullptr, 0, P);
    for (auto& w : writers) { try { w.get(); } catch (...) { if (!err) err = std::current_exception(); } }
    if (pinned) rg_host_free(pred);
    if (err) std::rethrow_exception(err);
  };

  auto level1_range = [&](rg_ctx* cx, int q0, int nq, bool write_out) {
    std::vector<double> cumsum((size_t)nq * NCS * R1), pred((size_t)nq * nchr * N);
    std::vector<int32_t> best(nq), converged(nq, 1);
    const double* tq = tau.data() + (size_t)q0 * R1;
    if (p.t2e) {   // one call per trait: the library derives the penalties from the score at beta = 0 and returns them
      rg_cox_options co{};
      co.niter_max = p.niter_max; co.niter_max_line_search = p.niter_max_line_search; co.niter_max_ridge = p.niter_max_ridge;
      co.niter_max_line_search_ridge = 100; co.numtol_cox = 2.5e-4; co.l1_ridge_tol = 1e-4;
      std::vector<double> tau_in(tau);   // --t2e-l1-pi6: the grid of the caller's (the call writes the penalties it used back into tau)
      for (int q = 0; q < nq; ++q) {
        co.tau = p.t2e_l1_pi6 ? tau_in.data() + (size_t)(q0 + q) * R1 : nullptr;
        double* cq = cumsum.data() + (size_t)q * NCS * R1;
        std::fill(cq, cq + (size_t)NCS * R1, 0.0);
        if (!r.pheno_pass[q0 + q]) { converged[q] = 0; best[q] = 0; continue; }   // no offset to fit against: skipped like the other trait modes
        check(cx, rg_l1_cox(cx, q0 + q, R1, r.Yraw.data() + (size_t)(q0 + q) * N, r.Yevent.data() + (size_t)(q0 + q) * N, r.offset.data() + (size_t)(q0 + q) * N, &co,
                            nchr, cols_per_chr.data(), tau.data() + (size_t)(q0 + q) * R1, cq + 5 * R1, &converged[q], &best[q], pred.data() + (size_t)q * nchr * N));
      }
    } else if (p.bt || p.ct) {
      rg_bt_options bo{};  // Regenie.hpp:287-290 defaults; family picks ridge_logistic_level_1* or ridge_poisson_level_1*
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
    const bool l1_batched = getenv("RG_L1_BATCHED") && atoi(getenv("RG_L1_BATCHED")) != 0;
    if (!l1_batched) l1_slots_start();
    if (!p.run_l1) {
      std::ostringstream lg;
      level0_range(ctx, 0, B, lg, pre_ring_ptr);
      sout << lg.str();
    }
    sout << "\n Level 1 ridge...\n";
    if (p.ct) sout << " Level 1 ridge with poisson regression...\n";
    if (p.t2e) sout << " Level 1 ridge with cox regression...\n";
    tl0 = std::chrono::steady_clock::now();
    if (l1_batched) level1_range(ctx, 0, P, true);   // all phenotypes in one call, then the files
    else level1_pipelined(ctx);
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
  // Every output file is written: the contexts' device memory (tens of GB of workspaces and W) and the runtime are left to process exit -- main()
  // leaves through _exit -- instead of being freed buffer by buffer (80 - 100 ms of a 0.4 s run at BASELINE configs[1]); RG_TEARDOWN=1 frees them.
  // INVARIANT the fast exit rests on: every output stream of the run (.loco / .prs / .firth / lists) is a local of a scope that has ended by
  // here, so it is flushed and closed; main() flushes the log and stdout itself.  A tool that finalises at exit (rocprofv3, sanitizers,
  // coverage) would lose its output, so the fast exit is off whenever one is detected, and RG_TEARDOWN=1 (the test suite sets it) always
  // takes the full path: contexts, RCCL communicators and the runtime are then torn down in order, which is also what surfaces leaks.
  if (full_teardown()) {
    if (grp) rg_group_destroy(grp);
    for (rg_ctx* cx : ctxs) rg_destroy(cx);
  } else fast_exit = true;
  sout << "\nElapsed time : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << "s\nEnd of run\n";
  return 0;
}

}  // namespace rgdrv
