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

__device__ __forceinline__ unsigned load_u8(const uint8_t* p, int64_t i, int64_t n) {
  return (i >= 0 && i < n) ? (unsigned)p[i] : 0xFFu;
}

// grid: (ceil(Np/16/256), n128, nblk); thread -> one output dword (16 positions) of one SNP row.
__global__ __launch_bounds__(256) void k_bed_prep_rows(const uint8_t* raw, int64_t raw_ld,
                                                       int64_t raw_blk_stride, uint8_t* pk,
                                                       int64_t pk_ld, int64_t pk_blk_stride,
                                                       const int32_t* d_bs, const uint8_t* act,
                                                       SegLayout seg, int64_t Np, int ref_first,
                                                       int64_t nfile_bytes,
                                                       int32_t* cnt_part /*[nblk][n128][2]*/,
                                                       uint8_t* pk4, int64_t pk4_ld, int64_t pk4_blk_stride) {
  const int blk = blockIdx.z, row = blockIdx.y;
  const int bs = d_bs[blk];
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;  // output dword index
  const int64_t nw = Np / 16;
  unsigned out = 0xFFFFFFFFu;
  int nmiss = 0, gsum = 0;
  if (w < nw && row < bs) {
    const int64_t pos = w * 16;
    // locate the fold segment (few segments: linear scan)
    int s = 0;
    for (int t = 1; t < seg.nseg; ++t)
      if (pos >= seg.pos_start[t]) s = t;
    const int64_t off = pos - seg.pos_start[s];
    int64_t nvalid = seg.len[s] - off;
    if (nvalid > 16) nvalid = 16;
    if (nvalid > 0) {
      const int64_t i0 = seg.file_start[s] + off;  // first file sample of this dword
      const uint8_t* r = raw + (int64_t)blk * raw_blk_stride + (int64_t)row * raw_ld;
      const int64_t b0 = i0 >> 2;
      const int sh = (int)(i0 & 3) * 2;
      unsigned long long v = 0;
#pragma unroll
      for (int t = 0; t < 5; ++t) v |= (unsigned long long)load_u8(r, b0 + t, nfile_bytes) << (8 * t);
      unsigned x = (unsigned)(v >> sh);
      if (ref_first) {  // swap 00 <-> 11, keep 01 (missing) and 10 (het)
        const unsigned lo = x & 0x55555555u, hi = (x >> 1) & 0x55555555u;
        const unsigned eq = ~(lo ^ hi) & 0x55555555u;
        x ^= eq | (eq << 1);
      }
      const unsigned a = *reinterpret_cast<const unsigned*>(act + w * 4);  // 11 per analysed sample
      unsigned vm = (nvalid >= 16) ? 0xFFFFFFFFu : ((1u << (2 * nvalid)) - 1u);
      const unsigned keep = a & vm;
      out = (x & keep) | ~keep;
      const unsigned lo = out & 0x55555555u, hi = (out >> 1) & 0x55555555u;
      const unsigned miss = lo & ~hi & 0x55555555u;
      const unsigned nlo = ~lo & 0x55555555u;
      nmiss = __popc(miss);
      gsum = 2 * __popc(nlo & ~hi) + __popc(nlo & hi);
    }
  }
  if (w < nw) *reinterpret_cast<unsigned*>(pk + (int64_t)blk * pk_blk_stride + (int64_t)row * pk_ld + w * 4) = out;
  if (pk4 && w < nw) {
    // FP4 E2M1 plane for the matrix cores (gram_fp4.hip): dosage 2 (code 00) -> 0100, 1 (code 10) -> 0010,
    // 0 / missing -> 0000; sample i of this dword -> nibble i of the 8 output bytes
    const unsigned lo = out & 0x55555555u, hi = (out >> 1) & 0x55555555u;
    const unsigned two = ~lo & ~hi & 0x55555555u, one = ~lo & hi;
    uint2 o;
    o.x = (spread8(two) << 2) | (spread8(one) << 1);
    o.y = (spread8(two >> 16) << 2) | (spread8(one >> 16) << 1);
    *reinterpret_cast<uint2*>(pk4 + (int64_t)blk * pk4_blk_stride + (int64_t)row * pk4_ld + w * 8) = o;
  }
  // block reduction of (nmiss, gsum) -> integer atomics (exact, order independent)
  for (int o = 32; o > 0; o >>= 1) {
    nmiss += __shfl_down(nmiss, o);
    gsum += __shfl_down(gsum, o);
  }
  if ((threadIdx.x & 63) == 0 && (nmiss | gsum)) {
    int32_t* c = cnt_part + ((int64_t)blk * gridDim.y + row) * 2;
    atomicAdd(c, nmiss);
    atomicAdd(c + 1, gsum);
  }
}

__global__ void k_bed_mu(const int32_t* cnt_part, const int32_t* d_bs, int n128, int n_active,
                         double* mu, int32_t* nmiss_blk) {
  const int blk = blockIdx.y;
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n128) return;
  const int32_t* c = cnt_part + ((int64_t)blk * n128 + row) * 2;
  const int bs = d_bs[blk];
  double m = 0.0;
  if (row < bs) {
    const int ns = n_active - c[0];
    m = (double)c[1] / (double)ns;  // total /= ns (Geno.cpp:1756); ns==0 -> inf/nan as reference
    if (c[0] > 0) atomicAdd(nmiss_blk + blk, c[0]);
  }
  mu[(int64_t)blk * n128 + row] = m;
}

void rg_launch_bed_prep(hipStream_t st, const uint8_t* raw, int64_t raw_ld, int64_t raw_blk_stride,
                        uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride, const int32_t* d_bs,
                        int nblk, int n128, const uint8_t* act, SegLayout seg, int64_t Np,
                        int ref_first, int n_active, double* mu, int32_t* nmiss, uint8_t* pk4,
                        int64_t pk4_ld, int64_t pk4_blk_stride) {
  // mu buffer is followed by an int32 scratch [nblk][n128][2] owned by the caller (see rg_ctx: the
  // scratch lives right behind d_mu); here we only receive pointers.
  int32_t* cnt = reinterpret_cast<int32_t*>(mu + (int64_t)nblk * n128);
  hipMemsetAsync(cnt, 0, sizeof(int32_t) * 2 * (size_t)nblk * n128, st);
  hipMemsetAsync(nmiss, 0, sizeof(int32_t) * nblk, st);
  const int64_t nw = Np / 16;
  dim3 grid((unsigned)((nw + 255) / 256), n128, nblk);
  hipLaunchKernelGGL(k_bed_prep_rows, grid, dim3(256), 0, st, raw, raw_ld, raw_blk_stride, pk, pk_ld,
                     pk_blk_stride, d_bs, act, seg, Np, ref_first, raw_ld, cnt, pk4, pk4_ld, pk4_blk_stride);
  hipLaunchKernelGGL(k_bed_mu, dim3((n128 + 127) / 128, nblk), dim3(128), 0, st, cnt, d_bs, n128,
                     n_active, mu, nmiss);
}

// ---------------------------------------------------------------------------------------------
// k_geno_xy: thread = one SNP row; workgroup = 256 rows x one chunk of positions (<= 4096, inside
// one fold).  Per 256-position sub-chunk the packed tile (256 rows x 64 bytes) and the V tile
// (CG columns x 256 positions) are staged through LDS with coalesced loads; V columns are processed
// in groups of CG to keep the accumulators in registers.
// part layout: [blk][chunk][row][2][Cv]   (0: sum g0*V, 1: sum miss*V)
#define CG 4
#define XP 80  // LDS row pitch of the packed tile: 64 data bytes + 16 (conflict-free 16-byte row reads)
__global__ __launch_bounds__(256) void k_geno_xy(const uint8_t* pk, int64_t pk_ld,
                                                 int64_t pk_blk_stride, const int32_t* d_bs, int n128,
                                                 const double* V, int64_t Np, int Cv,
                                                 const int64_t* chunk_pos, const int64_t* chunk_len,
                                                 int nchunk, double* part) {
  __shared__ double sV[CG][256];
  __shared__ __attribute__((aligned(16))) uint8_t sP[256 * XP];
  const int blk = blockIdx.z, ch = blockIdx.y;
  const int row0 = blockIdx.x * 256;
  const int row = row0 + threadIdx.x;
  const int bs = d_bs[blk];
  const int64_t p0 = chunk_pos[ch], plen = chunk_len[ch];
  const uint8_t* base = pk + (int64_t)blk * pk_blk_stride + p0 / 4;
  double* outp = part + ((((int64_t)blk * nchunk + ch) * n128 + row) * 2) * Cv;
  const bool live = row < bs;
  for (int c0 = 0; c0 < Cv; c0 += CG) {
    double a0[CG], am[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) a0[c] = am[c] = 0.0;
    for (int64_t q = 0; q < plen; q += 256) {
      const int npos = (int)min((int64_t)256, plen - q);  // multiple of 64
      __syncthreads();
      for (int t = threadIdx.x; t < CG * 256; t += 256) {
        const int c = t >> 8, i = t & 255;
        sV[c][i] = (c0 + c < Cv && i < npos) ? V[(int64_t)(c0 + c) * Np + p0 + q + i] : 0.0;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 64 + (threadIdx.x >> 2), piece = (threadIdx.x & 3) * 16;
        uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (row0 + r < bs && piece * 4 < npos)
          v = *reinterpret_cast<const uint4*>(base + (int64_t)(row0 + r) * pk_ld + q / 4 + piece);
        *reinterpret_cast<uint4*>(sP + r * XP + piece) = v;
      }
      __syncthreads();
      if (live) {
#pragma unroll 1
        for (int d = 0; d < 16; ++d) {
          const unsigned w = *reinterpret_cast<const unsigned*>(sP + threadIdx.x * XP + d * 4);
          if (w == 0xFFFFFFFFu) continue;  // 16 samples with dosage 0
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const unsigned code = (w >> (2 * i)) & 3u;
            const double g = (code == 0u) ? 2.0 : ((code == 2u) ? 1.0 : 0.0);
#pragma unroll
            for (int c = 0; c < CG; ++c) a0[c] = fma(g, sV[c][d * 16 + i], a0[c]);
            if (code == 1u) {
#pragma unroll
              for (int c = 0; c < CG; ++c) am[c] += sV[c][d * 16 + i];
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
                       const int32_t* d_bs, int nblk, int n128, const double* V, int64_t Np, int Cv,
                       const int64_t* chunk_pos, const int64_t* chunk_len, int nchunk, double* part) {
  dim3 grid((n128 + 255) / 256, nchunk, nblk);
  hipLaunchKernelGGL(k_geno_xy, grid, dim3(256), 0, st, pk, pk_ld, pk_blk_stride, d_bs, n128, V, Np,
                     Cv, chunk_pos, chunk_len, nchunk, part);
}
