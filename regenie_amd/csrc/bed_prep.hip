// Level-0 ingest kernels (HBM-bound byte work on 2-bit packed genotypes).
//
// k_bed_prep  : raw .bed rows (file sample order) -> "cleaned" packed rows in the fold-aligned
//               position space + per-SNP mean over analysed non-missing samples.
//               Stands in for the decode / mean / impute loop of readChunkFromBedFileToG
//               (reference src/Geno.cpp:1724-1763) and the `G *= ind_in_analysis` of
//               residualize_genotypes (src/Data.cpp:196): samples outside the analysis are
//               rewritten to code 11 (dosage 0, not missing), --ref-first swaps codes 00<->11
//               (Geno.cpp:1746), padding between folds is code 11.
// k_geno_xy   : per (chunk of positions, SNP): sum_i g0_ji V_i and sum_i miss_ji V_i for the
//               C + P columns V = [X | Y]; these give G~X and G~Y (the two skinny products of
//               Data.cpp:199 and Data.cpp:746) once combined with the SNP mean.
#include "rg_internal.h"

// flags at bit positions 0,2,..,14 (8 samples) -> bit 0 of nibbles 0..7
__device__ __forceinline__ unsigned spread8(unsigned y) {
  y &= 0x5555u;
  y = (y | (y << 8)) & 0x00FF00FFu;
  y = (y | (y << 4)) & 0x0F0F0F0Fu;
  y = (y | (y << 2)) & 0x11111111u;
  return y;
}

// grid: (n128, nblk); one workgroup per SNP row, thread -> groups of 64 positions (4 output dwords) g = tid, tid+256, ...
// The window of a group starts at an arbitrary 2-bit offset of the raw row (folds are re-aligned to 256 positions): it is
// cut out of up to five ALIGNED raw dwords.  Row totals (missing calls, dosage sum) are reduced inside the
// workgroup -- integer, hence exact and order independent -- and the SNP mean is written by the same launch.
__global__ __launch_bounds__(256) void k_bed_prep_rows(const uint8_t* const* __restrict__ rawptr, int64_t raw_ld,
                                                       uint8_t* __restrict__ pk,
                                                       int64_t pk_ld, int64_t pk_blk_stride,
                                                       const int32_t* __restrict__ d_bs,
                                                       const uint8_t* __restrict__ act, SegLayout seg, int64_t Np,
                                                       int ref_first, int n_active, double* __restrict__ mu,
                                                       int32_t* __restrict__ nmiss_blk, uint8_t* __restrict__ pk4,
                                                       int64_t pk4_ld, int64_t pk4_blk_stride) {
  __shared__ int red[4][2];
  const int blk = blockIdx.y, row = blockIdx.x;
  const int n128 = gridDim.x;
  const int bs = d_bs[blk];
  const int64_t nw = Np / 16;
  // raw rows may sit in the caller's own device buffer at any byte alignment: the window is cut out of the
  // aligned dwords around it (a dword that holds at least one valid byte never crosses into an unmapped page)
  const uint8_t* rowp = rawptr[blk] + (int64_t)min(row, bs - 1) * raw_ld;   // padding rows (>= bs) read nothing new
  uint32_t* po = reinterpret_cast<uint32_t*>(pk + (int64_t)blk * pk_blk_stride + (int64_t)row * pk_ld);
  uint2* p4 = pk4 ? reinterpret_cast<uint2*>(pk4 + (int64_t)blk * pk4_blk_stride + (int64_t)row * pk4_ld) : nullptr;
  const uint4* a128 = reinterpret_cast<const uint4*>(act);
  int nmiss = 0, gsum = 0;
  // A thread handles GROUPS of 64 positions (4 output dwords: one 16-byte store, two for the FP4 plane, one 16-byte load
  // of the activity bits) -- a quarter of the memory instructions of a dword-per-thread loop.  Fold segments start at
  // multiples of 256 positions, so a group never straddles one.  The 128 raw bits of a group start at an arbitrary 2-bit
  // offset of the row and are cut out of up to five ALIGNED raw dwords; dword j is fetched only if the window reaches it
  // (re-reading dword 0 otherwise), so no load ever leaves the row.  All loads of GI groups are issued (unconditionally,
  // at clamped addresses) before any is consumed.
  const bool row_live = row < bs;
  const int64_t ng = Np / 64;
  constexpr int GI = 2;
  for (int64_t g0 = threadIdx.x; g0 < ng; g0 += 256 * GI) {
    unsigned rw[GI][5];
    uint4 av[GI];
    int shv[GI], nv[GI];
#pragma unroll
    for (int u = 0; u < GI; ++u) {
      const int64_t gq = g0 + 256 * u;
      const int64_t gc = gq < ng ? gq : 0;
      const int64_t pos = gc * 64;
      int sgm = 0;
      for (int t = 1; t < seg.nseg; ++t)
        if (pos >= seg.pos_start[t]) sgm = t;
      const int64_t off = pos - seg.pos_start[sgm];
      int64_t nvalid = seg.len[sgm] - off;
      if (nvalid > 64) nvalid = 64;
      if (nvalid < 0 || gq >= ng || !row_live) nvalid = 0;
      const int64_t i0 = nvalid > 0 ? seg.file_start[sgm] + off : 0;   // first file sample of this group
      const uintptr_t ab = reinterpret_cast<uintptr_t>(rowp) + (uintptr_t)(i0 >> 2);   // byte holding sample i0
      const uint32_t* ap = reinterpret_cast<const uint32_t*>(ab & ~(uintptr_t)3);
      const int sh = (int)(ab & 3) * 8 + (int)(i0 & 3) * 2;                            // <= 30
      const int nbits = sh + 2 * (int)nvalid;                                          // window end, in bits from ap[0]
#pragma unroll
      for (int j = 0; j < 5; ++j) rw[u][j] = ap[(j == 0 || nbits > 32 * j) ? j : 0];
      av[u] = a128[gc];                                    // 11 per analysed sample
      shv[u] = sh;
      nv[u] = (int)nvalid;
    }
#pragma unroll
    for (int u = 0; u < GI; ++u) {
      const int64_t gq = g0 + 256 * u;
      if (gq >= ng) continue;
      unsigned outw[4];
      const unsigned avw[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        unsigned out = 0xFFFFFFFFu;
        const int nvd = nv[u] - 16 * d;                    // valid samples of this dword
        if (nvd > 0) {
          unsigned x = (unsigned)((((unsigned long long)rw[u][d + 1] << 32) | rw[u][d]) >> shv[u]);
          if (ref_first) {  // swap 00 <-> 11, keep 01 (missing) and 10 (het)
            const unsigned lo = x & 0x55555555u, hi = (x >> 1) & 0x55555555u;
            const unsigned eq = ~(lo ^ hi) & 0x55555555u;
            x ^= eq | (eq << 1);
          }
          const unsigned vm = (nvd >= 16) ? 0xFFFFFFFFu : ((1u << (2 * nvd)) - 1u);
          const unsigned keep = avw[d] & vm;
          out = (x & keep) | ~keep;
          const unsigned lo = out & 0x55555555u, hi = (out >> 1) & 0x55555555u;
          const unsigned nlo = ~lo & 0x55555555u;
          nmiss += __popc(lo & ~hi & 0x55555555u);
          gsum += 2 * __popc(nlo & ~hi) + __popc(nlo & hi);
        }
        outw[d] = out;
      }
      reinterpret_cast<uint4*>(po)[gq] = make_uint4(outw[0], outw[1], outw[2], outw[3]);
      if (p4) {
        // FP4 E2M1 plane for the matrix cores (gram_fp4.hip): dosage 2 (code 00) -> 0100, 1 (code 10) -> 0010,
        // 0 / missing -> 0000; sample i of a dword -> nibble i of its 8 output bytes
        unsigned o8[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned lo = outw[d] & 0x55555555u, hi = (outw[d] >> 1) & 0x55555555u;
          const unsigned two = ~lo & ~hi & 0x55555555u, one = ~lo & hi;
          o8[2 * d] = (spread8(two) << 2) | (spread8(one) << 1);
          o8[2 * d + 1] = (spread8(two >> 16) << 2) | (spread8(one >> 16) << 1);
        }
        uint4* q4 = reinterpret_cast<uint4*>(p4) + 2 * gq;
        q4[0] = make_uint4(o8[0], o8[1], o8[2], o8[3]);
        q4[1] = make_uint4(o8[4], o8[5], o8[6], o8[7]);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    nmiss += __shfl_down(nmiss, o);
    gsum += __shfl_down(gsum, o);
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = nmiss; red[threadIdx.x >> 6][1] = gsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nm = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    const int gs = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    double m = 0.0;
    if (row < bs) {
      m = (double)gs / (double)(n_active - nm);   // total /= ns (Geno.cpp:1756); ns == 0 -> inf/nan as the reference
      if (nm > 0) atomicAdd(nmiss_blk + blk, nm);
    }
    mu[(int64_t)blk * n128 + row] = m;
  }
}

void rg_launch_bed_prep(hipStream_t st, const uint8_t* const* rawptr, int64_t raw_ld,
                        uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride, const int32_t* d_bs,
                        int nblk, int n128, const uint8_t* act, SegLayout seg, int64_t Np,
                        int ref_first, int n_active, double* mu, int32_t* nmiss, uint8_t* pk4,
                        int64_t pk4_ld, int64_t pk4_blk_stride) {
  hipMemsetAsync(nmiss, 0, sizeof(int32_t) * nblk, st);
  hipLaunchKernelGGL(k_bed_prep_rows, dim3(n128, nblk), dim3(256), 0, st, rawptr, raw_ld, pk, pk_ld,
                     pk_blk_stride, d_bs, act, seg, Np, ref_first, n_active, mu, nmiss, pk4, pk4_ld, pk4_blk_stride);
}

// ---------------------------------------------------------------------------------------------
// k_geno_xy: thread = one SNP row; workgroup = 256 rows x one chunk of positions (<= 4096, inside
// one fold).  Per 256-position sub-chunk the packed tile (256 rows x 64 bytes) is staged through LDS
// with coalesced loads.  The V = [X | Y] values of a position are the same for every row, i.e. uniform
// across the wave: they are read through the scalar cache (s_load) and enter the fp64 FMAs as SGPR
// operands, so the inner loop is decode (bit-field extract + convert, 2 ops) + CG FMAs per sample.
// The missing-indicator sums are only formed for blocks that have missing calls.
// part layout: [blk][chunk][row][2][Cv]   (0: sum g0*V, 1: sum miss*V)
#define XP 80  // LDS row pitch of the packed tile: 64 data bytes + 16 (conflict-free 16-byte row reads)
// CG = columns of V per pass over the packed rows (every pass re-stages and re-decodes them), SB = samples per batch of
// scalar loads (CG * SB doubles = 2 * CG * SB SGPRs): 4 x 8 for up to four columns, 8 x 4 beyond.
template <int CG, int SB>
__global__ __launch_bounds__(256) void k_geno_xy(const uint8_t* __restrict__ pk, int64_t pk_ld,
                                                 int64_t pk_blk_stride, const int32_t* __restrict__ d_bs,
                                                 const int32_t* __restrict__ nmiss_blk, int n128,
                                                 const double* __restrict__ V, int64_t Np, int Cv,
                                                 const int64_t* __restrict__ chunk_pos,
                                                 const int64_t* __restrict__ chunk_len, int nchunk,
                                                 double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) uint8_t sP[256 * XP];
  const int blk = blockIdx.z, ch = blockIdx.y;
  const int row0 = blockIdx.x * 256;
  const int row = row0 + threadIdx.x;
  const int bs = d_bs[blk];
  const bool has_miss = nmiss_blk[blk] > 0;
  const int64_t p0 = chunk_pos[ch], plen = chunk_len[ch];
  const uint8_t* base = pk + (int64_t)blk * pk_blk_stride + p0 / 4;
  double* outp = part + ((((int64_t)blk * nchunk + ch) * n128 + row) * 2) * Cv;
  const bool live = row < bs;
  for (int c0 = 0; c0 < Cv; c0 += CG) {
    double a0[CG], am[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) a0[c] = am[c] = 0.0;
    // columns beyond Cv re-read column Cv-1 (their sums are discarded): keeps every scalar load in bounds
    const double* Vc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) Vc[c] = V + (int64_t)min(c0 + c, Cv - 1) * Np + p0;
    for (int64_t q = 0; q < plen; q += 256) {
      const int npos = (int)min((int64_t)256, plen - q);  // multiple of 64
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 64 + (threadIdx.x >> 2), piece = (threadIdx.x & 3) * 16;
        // unconditional load at a clamped address + select (a load under `if` costs a full round trip each)
        const bool ok = (row0 + r < bs) && (piece * 4 < npos);
        uint4 v = *reinterpret_cast<const uint4*>(base + (int64_t)min(row0 + r, bs - 1) * pk_ld + q / 4 +
                                                  (piece * 4 < npos ? piece : 0));
        if (!ok) v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        *reinterpret_cast<uint4*>(sP + r * XP + piece) = v;
      }
      __syncthreads();
      const int nd = npos / 16;
#pragma unroll 1
      for (int d = 0; d < nd; ++d) {
        const unsigned w = *reinterpret_cast<const unsigned*>(sP + threadIdx.x * XP + d * 4);
        const unsigned lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
        const unsigned dd = (hi & ~lo) | ((~(hi | lo) & 0x55555555u) << 1);   // 2-bit dosage fields
        const unsigned ms = lo & ~hi;
        // SB samples at a time: SB x CG doubles of V = 64 SGPRs per scalar-load batch
#pragma unroll 1
        for (int hf = 0; hf < 16 / SB; ++hf) {
          const int64_t vo = q + d * 16 + hf * SB;
          const unsigned dh = dd >> (2 * SB * hf);
#pragma unroll
          for (int i = 0; i < SB; ++i) {
            const double g = (double)((dh >> (2 * i)) & 3u);
#pragma unroll
            for (int c = 0; c < CG; ++c) a0[c] = fma(g, Vc[c][vo + i], a0[c]);
          }
          if (has_miss) {
            const unsigned mh = ms >> (2 * SB * hf);
#pragma unroll
            for (int i = 0; i < SB; ++i) {
              const double m = (double)((mh >> (2 * i)) & 1u);
#pragma unroll
              for (int c = 0; c < CG; ++c) am[c] = fma(m, Vc[c][vo + i], am[c]);
            }
          }
        }
      }
    }
    if (row < n128) {
#pragma unroll
      for (int c = 0; c < CG; ++c)
        if (c0 + c < Cv) {
          outp[c0 + c] = live ? a0[c] : 0.0;
          outp[Cv + c0 + c] = live ? am[c] : 0.0;
        }
    }
  }
}

void rg_launch_geno_xy(hipStream_t st, const uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride,
                       const int32_t* d_bs, const int32_t* nmiss, int nblk, int n128, const double* V, int64_t Np,
                       int Cv, const int64_t* chunk_pos, const int64_t* chunk_len, int nchunk, double* part) {
  dim3 grid((n128 + 255) / 256, nchunk, nblk);
  if (Cv <= 4)
    hipLaunchKernelGGL((k_geno_xy<4, 8>), grid, dim3(256), 0, st, pk, pk_ld, pk_blk_stride, d_bs, nmiss, n128, V, Np, Cv,
                       chunk_pos, chunk_len, nchunk, part);
  else
    hipLaunchKernelGGL((k_geno_xy<8, 4>), grid, dim3(256), 0, st, pk, pk_ld, pk_blk_stride, d_bs, nmiss, n128, V, Np, Cv,
                       chunk_pos, chunk_len, nchunk, part);
}
