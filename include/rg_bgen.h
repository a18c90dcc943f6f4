/* rg_bgen.h -- C ABI of the BGEN v1.2 input of the Step-1 path (`regenie --step 1 --bgen FILE`).
 *
 * It replaces, for the files regenie reads through its own fast path (layout 2, 8-bit probabilities, unphased, biallelic,
 * diploid; zlib / zstd / no compression -- check_bgen, Geno.cpp:1826-1958):
 *   rg_bgen_open            the variant scan of prep_bgen (BgenParser::read_variant loop)   Geno.cpp:38-175
 *   rg_bgen_sample_id       BgenParser::get_sample_ids (embedded identifiers = FID_IID)     Geno.cpp:146-152
 *   rg_bgen_variant         snpinfo[] fields: chromosome, position, rsid, alleles, offset   Geno.cpp:73-128
 *   rg_bgen_read_dosages    readChunkFromBGEN + readChunkFromBGENFileToG_fast               Geno.cpp:2122-2171, :1574-1699
 *   rg_bgen_read_dosages_info   the Step-2 form: parseSnpfromBGEN's dosages and info-score terms   Geno.cpp:2186-2330
 *   rg_bgen_read_blocks     the inflate half of parseSnpfromBGEN (the caller walks the bytes)  Geno.cpp:2219-2262
 * The rows are ALT-count style dosages in [0, 2] (G = prob1 + 2 prob0, or prob1 + 2 prob2 with ref_first), -3 = missing:
 * the input of rg_l0_blocks_f64 (rg_step1.h), which applies the reference's mean imputation.  Host-only code.
 * Conventions as in rg_pgen.h: 0 on success, <0 on error, rg_bgen_last_error(h); rg_bgen_open always stores a handle.
 */
#ifndef RG_BGEN_H
#define RG_BGEN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_bgen rg_bgen;

#define RG_BGEN_OK 0
#define RG_BGEN_ERR_ARG (-1)
#define RG_BGEN_ERR_FORMAT (-2)      /* not a bgen file, malformed or truncated, inflate failure */
#define RG_BGEN_ERR_UNSUPPORTED (-3) /* layout 1, phased / non-8-bit / multiallelic / non-diploid data */
#define RG_BGEN_ERR_DEVICE (-4)      /* device path: no MI355X / HIP device, or a HIP call failed */

int rg_bgen_open(rg_bgen** out, const char* path); /* parses the header and scans the variant identifying data */
void rg_bgen_close(rg_bgen* h);
const char* rg_bgen_last_error(const rg_bgen* h);

/* compression: 0 none, 1 zlib, 2 zstd.  Any pointer may be NULL. */
int rg_bgen_info(const rg_bgen* h, int64_t* n_samples, int64_t* n_variants, int32_t* compression,
                 int32_t* has_sample_ids);
/* Embedded sample identifier i (regenie uses it as FID_IID as it is); the pointer stays valid until rg_bgen_close. */
int rg_bgen_sample_id(const rg_bgen* h, int64_t i, const char** id);
/* Identifying data of variant j (file order).  Strings stay valid until rg_bgen_close; file_offset is the position of the
 * record (regenie's snpinfo[].offset). */
int rg_bgen_variant(const rg_bgen* h, int64_t j, const char** chrom, uint32_t* position, const char** rsid,
                    const char** allele0, const char** allele1, int64_t* file_offset);
/* Worker threads of rg_bgen_read_dosages (default 1; the reference inflates a block's variants under OpenMP). */
int rg_bgen_set_threads(rg_bgen* h, int32_t n_threads);
/* Dosage rows of n variants: rows[k * row_stride .. + n_samples) doubles. */
int rg_bgen_read_dosages(rg_bgen* h, int64_t n, const int64_t* variant_idx, int32_t ref_first, double* rows,
                         int64_t row_stride);
/* The same, plus each sample's term of the IMPUTE info score the Step-2 reader accumulates (parseSnpfromBGEN, Geno.cpp:2292-2295:
 * 4 prob0 + prob1 - G^2, or with ref_first 4 prob2 + prob1 - G^2; 0 for a missing sample): info_rows has the layout of rows. */
int rg_bgen_read_dosages_info(rg_bgen* h, int64_t n, const int64_t* variant_idx, int32_t ref_first, double* rows, double* info_rows,
                              int64_t row_stride);
/* The inflated, checked probability blocks of n variants, for a caller that walks the bytes itself (the Step-2 driver turns them into
 * 2-byte integer dosages and the allele / info sums in ONE pass instead of three double rows per variant): block k starts at
 * blocks + k * block_stride (block_stride >= rg_bgen_block_bytes = 10 + 3 N) and holds, as the file does,
 *   [0,4) N   [4,6) K = 2   [6] [7] ploidy 2 2   [8, 8+N) ploidy byte per sample, bit 7 = missing   [8+N] phased = 0   [9+N] bits = 8
 *   [10+N, 10+3N) two probability bytes per sample (prob0, prob1, in units of 1 / 255).
 * Same checks and messages as rg_bgen_read_dosages; n_threads workers (0: as set by rg_bgen_set_threads), one variant at a time each.
 * The read calls do not modify the handle: several threads may read through one handle concurrently (rg_bgen_last_error returns the
 * calling thread's own last message). */
int rg_bgen_block_bytes(const rg_bgen* h, int64_t* bytes);
int rg_bgen_read_blocks(rg_bgen* h, int64_t n, const int64_t* variant_idx, uint8_t* blocks, int64_t block_stride, int32_t n_threads);


/* ---- device path (csrc/bgen_inflate.hip): inflate + walk of parseSnpfromBGEN (Geno.cpp:2186-2330) on the GPU ----------------------------
 * The host reads the STORED bytes only; the zlib streams of a batch of variants are inflated one per wavefront (k_bgen_inflate), checked
 * (header fields, Adler-32) and walked into uint16 dosage rows in units of 1 / 255 (0xFFFF = missing) -- the input of rg_s2_qt_block_int /
 * rg_s2_bt_score_int with g_on_device = 1 -- and the allele / info sums as exact integers.  zlib-compressed files only.
 *
 * rg_bgen_compressed_bytes   size of the buffer rg_bgen_read_compressed needs for these variants
 * rg_bgen_read_compressed    stream k at dst + off[k] (a multiple of 16), clen[k] bytes, inflating to ulen[k] bytes (the file's own field);
 *                            n_threads workers (0: as set by rg_bgen_set_threads).  The handle is only read. */
int rg_bgen_compressed_bytes(const rg_bgen* h, int64_t n, const int64_t* variant_idx, int64_t* bytes);
int rg_bgen_read_compressed(rg_bgen* h, int64_t n, const int64_t* variant_idx, uint8_t* dst, int64_t cap, int64_t* off, int32_t* clen,
                            int32_t* ulen, int32_t n_threads);

typedef struct rg_bgen_dev rg_bgen_dev;     /* a decoder on one GPU: its own stream, two slots of buffers (a batch can be decoded while the
                                             * previous one is being tested) */
int rg_bgen_dev_create(rg_bgen_dev** out, int32_t device);
void rg_bgen_dev_destroy(rg_bgen_dev* d);
const char* rg_bgen_dev_last_error(const rg_bgen_dev* d);
/* The samples of the rows: file_idx[k] = position in the file of analysed sample k (NULL: the n_file samples in file order, n = n_file);
 * mask: [P][n] bytes, 1 = trait p is observed for sample k (NULL or P = 0: no per-trait sums; at most 2,048 traits with masks, RG_BGEN_ERR_ARG beyond:
 * the caller then keeps the host route). */
int rg_bgen_dev_set_samples(rg_bgen_dev* d, int64_t n_file, int64_t n, const int64_t* file_idx, int32_t P, const uint8_t* mask);
/* What a batch leaves behind.  The host arrays may be NULL (not wanted); g16 / raw are DEVICE pointers owned by the decoder, valid until the
 * next call on the same slot.  Sums run over the analysed samples whose genotype is not missing:
 *   sum_q = sum q_i (q = b1 + 2 b0, or b1 + 2 max(255 - b0 - b1, 0) with ref_first: the dosage x 255);
 *   sum_info = sum 255 (4 bX + b1) - q^2 (the IMPUTE-info numerator x 65025);  n_obs;  max_q (above 510: prob0 + prob1 > 1 in the file);
 *   *_t [nvar][P]: the same over the samples MISSING for trait p (what parseSnpfromBGEN's per-trait sums leave out);
 *   status: 0 = the stream inflated to the stated size, its Adler-32 matches and the block is layout 2 / 2 alleles / ploidy 2 / unphased /
 *   8 bits; anything else (the values are the decoder's own) marks a variant the caller must read through the host route. */
typedef struct rg_bgen_dev_out {
  const uint16_t* g16; int64_t ld16;
  const uint8_t* raw; int64_t raw_stride;
  int64_t* sum_q; int64_t* sum_info; int64_t* n_obs; int32_t* max_q;
  int64_t* sum_q_t; int64_t* sum_info_t; int64_t* n_obs_t;
  int32_t* status;
} rg_bgen_dev_out;
/* comp: host buffer holding the streams (as rg_bgen_read_compressed left them; page-locked memory makes the copy asynchronous);
 * blocks until the batch is decoded. */
int rg_bgen_dev_decode(rg_bgen_dev* d, int32_t slot, int32_t nvar, const uint8_t* comp, int64_t comp_bytes, const int64_t* off,
                       const int32_t* clen, const int32_t* ulen, int32_t ref_first, rg_bgen_dev_out* out);
/* copies n bytes from a device pointer of rg_bgen_dev_out to the host (tests: the inflated blocks, the dosage rows) */
int rg_bgen_dev_fetch(rg_bgen_dev* d, const void* device_ptr, void* dst, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
