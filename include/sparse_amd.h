/*
 * sparse_amd.h — C ABI of libsparse_amd.so, the MI355X (gfx950) hot path behind the
 * pydata/sparse `numba_backend` dot / elementwise / reduction kernels.
 *
 * Boundary contract (mirrors the reference's kernel-factory boundary):
 *   - The reference exposes Python-callable kernels over *raw arrays*, keyed by a dtype
 *     pair (`_dot_csr_ndarray_type(dt1, dt2)(out_shape, a_data, a_indices, a_indptr, b)`,
 *     sparse/numba_backend/_common.py:720-755).  Here the dtype key is an explicit
 *     `val_dtype` / `idx_dtype` code and every array is a plain device pointer.
 *   - The CALLER owns every buffer (inputs, outputs, workspace).  Nothing is allocated
 *     behind the caller's back; data-dependent output sizes use the reference's own
 *     two-phase shape (count -> scan -> fill) as two entry points.
 *   - All pointers are DEVICE pointers (HBM) unless a parameter says "host".
 *   - Every entry point is asynchronous on `stream` (a hipStream_t passed as void*),
 *     stateless, re-entrant and thread-safe; it returns 0 on success, a positive
 *     hipError_t value if the HIP runtime reported one, or a negative SPAMD_E* code for
 *     an argument error.  It never throws and never synchronises the device.
 *   - No torch / C++ types cross this boundary.
 */
#ifndef SPARSE_AMD_H
#define SPARSE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dtype codes (shared by value and index arguments) */
#define SPAMD_F32 0
#define SPAMD_F64 1
#define SPAMD_I32 2
#define SPAMD_I64 3
#define SPAMD_BF16 4
#define SPAMD_U8 5 /* bool (0/1) — results of comparisons, any/all, astype(bool) */

#define SPAMD_MAX_NDIM 16 /* largest array rank the key kernels accept */

/* error codes (negative; positive values are hipError_t) */
#define SPAMD_EINVAL (-1)  /* bad size / null pointer / misaligned */
#define SPAMD_ETYPE (-2)   /* unsupported dtype combination */
#define SPAMD_EWS (-3)     /* workspace too small */

/* flags */
#define SPAMD_TILED_GROUP_ENDS 2u /* spamd_spmm_tiled: blk_off has tiles + 1 entries per row group (spamd_spmm_tiled_inspect) */
#define SPAMD_EXACT_MULADD 1u /* separate IEEE mul + add (bit-exact vs the reference's
                                 non-contracted loop) instead of fused multiply-add */
#define SPAMD_TILED_INT32 8u /* spamd_spmm_tiled with val_dtype SPAMD_F32: the stream's values, B and the result are int32 BIT
                             * PATTERNS (the inspector only moves value bits: build the stream from the int32 values viewed
                             * as float32); products and sums wrap around like NumPy's int32 (`_dot_dtype`, _common.py:635) */
#define SPAMD_SPMM_ROWVEC 16u /* spamd_spmm_csr: results of at most 4 columns keep the row-vector kernel (lanes along a row) instead of the stream form */
#define SPAMD_SPMM_ROWGROUP 4u /* spamd_spmm_csr: the k-ascending row-group kernel whatever the shape (no row-vector, no LDS-resident-B path) */

/* Library/ABI version: major*10000 + minor*100 + patch. */
int spamd_version(void);

/* Name of the code object's target ("gfx950"). Host pointer to a static string. */
const char* spamd_target_arch(void);

/* ---------------------------------------------------------------------------------------
 * A1  CSR x dense -> dense        replaces `_dot_csr_ndarray`
 *                                  (sparse/numba_backend/_common.py:720-755; dispatch :386-389)
 *   out[i, j] = sum_{k in row i, ascending storage order} a_data[k] * b[a_indices[k], j]
 *   A is M x K in CSR (a_indptr has M+1 entries), B is K x N row-major with leading
 *   dimension ldb (elements), out is M x N row-major with leading dimension ldo.
 *   Every out element is WRITTEN (rows without stored elements get zeros): the caller
 *   does not pre-zero `out`.  Accumulation is in the value dtype (the reference's
 *   `dtr = dt1*dt2`; the host layer promotes mixed inputs), strictly in storage order
 *   per output element, so with SPAMD_EXACT_MULADD the result is bit-identical to the
 *   reference loop; without it each step is one fused multiply-add.
 *   Results of at most 4 columns (N = 1: the matrix-vector product) without SPAMD_EXACT_MULADD take the row-vector
 *   kernel instead: the lanes of a wave run along a row and a butterfly adds their partial sums, i.e. a fixed TREE
 *   order per row (deterministic; floating point within rounding of the k-ascending sum, integers identical).
 *   SPAMD_SPMM_ROWGROUP keeps the k-ascending kernel for those widths as well.
 *   val_dtype: F32 | F64 | I32 | I64.   idx_dtype: I32 | I64 (a_indices and a_indptr).
 * ------------------------------------------------------------------------------------- */
int spamd_spmm_csr(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N,
                   const void* a_data, const void* a_indices, const void* a_indptr,
                   const void* b, int64_t ldb, void* out, int64_t ldo,
                   unsigned flags, void* stream);

/* The same product with B RESIDENT IN LDS, for a short contracted axis (`_dot_csr_ndarray` / `_dot_coo_ndarray` as
 * tensordot uses them, `_common.py:720-755, 979-1014`; BASELINE config 3: K = 512): a 256-byte column panel of B —
 * (K + 1) * 256 bytes <= 144 KB, i.e. K <= 575 — is copied into LDS once per workgroup, 16 lanes own a row of A and read a
 * row of B per stored element from LDS instead of through the vector-memory path.  Sums run in storage order per output
 * element (bit-identical to spamd_spmm_csr's row-group kernel in both arithmetic modes; SPAMD_EXACT_MULADD as there).
 * spamd_spmm_csr itself takes this path when `spamd_spmm_csr_ldsb_fits` says 1, M >= 8192 and N * itemsize >= 128, unless
 * SPAMD_SPMM_ROWGROUP is set.  `..._fits`: K within the LDS budget, N / ldb / ldo multiples of 16 / itemsize, b and out
 * 16-byte aligned. */
int spamd_spmm_csr_ldsb_fits(int val_dtype, int64_t M, int64_t K, int64_t N, const void* b, int64_t ldb,
                             const void* out, int64_t ldo);
int spamd_spmm_csr_ldsb(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N,
                        const void* a_data, const void* a_indices, const void* a_indptr,
                        const void* b, int64_t ldb, void* out, int64_t ldo,
                        unsigned flags, void* stream);

/* The same product for results of at most 4 columns (N = 1: `x @ v`, the matrix-vector product) organised around the
 * CSR triplet's STREAM (`_dot_csr_ndarray`, `_common.py:744-753`, whose inner loop over one output column is this product):
 * every wave owns a piece of the stream that starts and ends on row boundaries, reads it once in 16-byte loads with four
 * 256-element subtiles in flight, takes B from LDS (K * N values <= 160 KB - 512 B), closes the rows with a segmented scan
 * over the lanes and stores `out` in row order.  A row's products are added in a fixed tree order (deterministic; floating
 * point within rounding of the storage-order sum, integers identical) - SPAMD_EXACT_MULADD is not accepted here.
 * spamd_spmm_csr takes this path for N <= 4 when `..._fits` says 1 and M >= 32768, unless SPAMD_SPMM_ROWGROUP,
 * SPAMD_SPMM_ROWVEC or (for floating point) SPAMD_EXACT_MULADD is set.  `..._fits`: N in 1..4 (1..3 for 8-byte values), M < 2^28, B within the LDS
 * budget, a_data and a_indices 16-byte aligned.  `nnz` = a_indptr[M] when the caller knows it, -1 otherwise (the kernel then reads it: one
 * more memory latency at the head of every wave).  flags bits 8..15 (tuning hint, 0 = default): workgroups per resident slot.
 * Round 6: `..._fits` returns the number of PASSES over A the product takes - ceil(N / w) for N up to 64 with w = the widest
 * chunk of columns (<= 4, 8-byte values: 3) whose K x w values fit the LDS; 0 = not this kernel - and spamd_spmm_csr_stream
 * runs them, one launch per chunk (b and out may be strided: ldb, ldo).  A pass costs A's stream whatever its width, so
 * spamd_spmm_csr takes up to 3 passes for 5 <= N <= SPAMD_STREAM_MULTI_MAX_N, up to 2 for N <= 4 columns of 8-byte values
 * and 1 for N <= 4 columns of 4-byte values, under the conditions above. */
#define SPAMD_STREAM_MULTI_MAX_N 12
int spamd_spmm_csr_stream_fits(int val_dtype, int64_t M, int64_t K, int64_t N, const void* a_data, const void* a_indices);
int spamd_spmm_csr_stream(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N,
                          const void* a_data, const void* a_indices, const void* a_indptr,
                          const void* b, int64_t ldb, void* out, int64_t ldo,
                          int64_t nnz, unsigned flags, void* stream);

/* ---------------------------------------------------------------------------------------
 * A1 (inspector/executor form)   same product as spamd_spmm_csr for F32 (N % 128 == 0) and F64 (N % 64 == 0),
 *   both arithmetic modes, with a cached K-tiled copy of A — the role `cusparseSpMM_preprocess`-style
 *   inspection plays elsewhere; the reference's analogue is the memoised format conversion of
 *   `COO(cache=True)` (sparse/numba_backend/_coo/core.py:317-338).  Inspector (once per matrix):
 *     keys   = spamd_csr_to_keys(A)                      row*K + col
 *     tkeys  = spamd_spmm_tiled_keys(keys)               ((g*ntiles+t)*RG + row%RG)*KB + col%KB
 *     stable sort of (tkeys, values)                     spamd_sort_kv
 *     seg_start, nblk = spamd_spmm_tiled_lists(tkeys)    first element / blocks of list (g,t)
 *     blk_off = spamd_exclusive_scan(nblk)               int64[nseg + 1], nseg = groups*ntiles
 *     blk_off32 = spamd_convert(I64 -> I32)              what the executor reads (total blocks < 2^31)
 *     blocks  = spamd_spmm_tiled_pack(...)               64-byte blocks, d0 = col%KB << 9 | 2 + 2*(row%RG):
 *                                                        F32: eight (d0, value bits);
 *                                                        F64: five d0, one pad dword, five (value lo, value hi);
 *                                                        lists padded with zero entries
 *   where groups = ceil(ceil(M/RG) / GPB) * GPB (every wave of the executor grid has lists).
 *   Executor: spamd_spmm_tiled — a workgroup covers a panel of 512 bytes of the output rows (128 F32 or 64 F64
 *   columns; wider N = more panels); B streams through LDS in KB-row tiles (LDS-DMA), each wave pulls its blocks
 *   with scalar loads and keeps 32 rows of partial sums in a fixed VGPR block addressed with s_set_gpr_idx.
 *   RG / KB / GPB / entries per block / slack / panel width from spamd_spmm_tiled_params(val_dtype);
 *   `blocks` must hold total_blocks + slack blocks and be 64-byte aligned.
 *   flags: SPAMD_EXACT_MULADD as for spamd_spmm_csr; bits 8..15 (optional tuning hint, 0 = default): 64-byte lines of
 *   each list to prefetch into L2 two tile phases ahead (~1.5 x the mean blocks per list).  Results are bit-identical to spamd_spmm_csr with the
 *   same flags (sorted column indices), hence to the reference loop under SPAMD_EXACT_MULADD.
 * ------------------------------------------------------------------------------------- */
int spamd_spmm_tiled_params(int val_dtype, int* rows_per_group, int* tile_rows, int* groups_per_block,
                            int* entries_per_block, int* slack_blocks, int* direct_max_tiles, int* panel_cols);
/* Direct (sort-free) inspector for CSR with sorted column indices and ceil(K/KB) <= direct_max_tiles: the tiled
 * order is a stable partition by tile of every RG-row group of the CSR order.
 *   spamd_spmm_tiled_count: nblk[nseg+1] (blocks per list, ready for spamd_exclusive_scan), flags[0] = 1 if a row's
 *                           column indices are not ascending (then use the key-sort recipe above);
 *   spamd_spmm_tiled_fill:  writes the block stream from (a_data of val_dtype, a_indices, a_indptr) and blk_off. */
int spamd_spmm_tiled_count(int val_dtype, int idx_dtype, int64_t M, int64_t K, const void* a_indices,
                           const void* a_indptr, int64_t* nblk, int* flags, void* stream);
int spamd_spmm_tiled_fill(int val_dtype, int idx_dtype, int64_t M, int64_t K, const void* a_data, const void* a_indices,
                          const void* a_indptr, const int64_t* blk_off, int64_t total_blocks, int* blocks,
                          void* stream);
/* one-pass form of count + fill: every row group's first block follows from the row pointers alone (an upper bound of what
 * the groups before it need: no scan, no dependence between groups), so blk_off is int32[groups * (tiles + 1)] — per group
 * the first block of each of its lists and the end of the last one; pass SPAMD_TILED_GROUP_ENDS to spamd_spmm_tiled with
 * it.  `blocks` has room for ceil(nnz / entries_per_block) + lists + slack_blocks blocks (the blocks between groups are
 * zeroed); state = one 64-bit word; state[0] != 0 afterwards: unsorted column indices, outputs invalid (use the key-sort
 * recipe) */
int spamd_spmm_tiled_inspect(int val_dtype, int idx_dtype, int64_t M, int64_t K, const void* a_data, const void* a_indices,
                             const void* a_indptr, void* state, int* blk_off, int* blocks, void* stream);
/* The same outputs straight from a CSC operand (a_indices = row indices, a_indptr = K + 1 column pointers, rows ascending
 * inside every column) - no CSC -> CSR conversion: a K-tile is 160 whole columns of the CSC arrays and a workgroup's rows own
 * one short run per column.  ws = int32 workspace of spamd_spmm_tiled_inspect_csc_ws(M, K) words, 8-byte aligned;
 * state[0] != 0 afterwards: rows out of order - the lists are then EMPTY (blk_off all zero) and the caller converts to CSR.  Replaces, for `csc @ dense`, the reference's `_dot_csc_ndarray`
 * (_common.py:869-904) together with the conversion this backend needed before it. */
int64_t spamd_spmm_tiled_inspect_csc_ws(int64_t M, int64_t K);
int spamd_spmm_tiled_inspect_csc(int val_dtype, int idx_dtype, int64_t M, int64_t K, const void* a_data,
                                 const void* a_indices, const void* a_indptr, int* ws, void* state, int* blk_off,
                                 int* blocks, void* stream);
/* Balanced layouts for SKEWED matrices (round 5; the reference's loop, _common.py:744-753, is indifferent to row lengths -
 * one wave per 35 consecutive rows is not: Zipf row lengths cost 4x at config 2's size).  Rows are sorted by length and classed
 * against `cap` (a power of two; longer than cap/2: a group of its own, then 2 / 4 / 8 / 16 rows to a group, everything else 35),
 * so that the 16 groups of a workgroup are equally heavy; rows are read and stored through a row map; entries, lists and the
 * order of every output element's terms are unchanged (bit-identical products).
 *   spamd_spmm_tiled_map_stats   keys[M] = K - length of every row, rows[M] = 0..M-1, stats[8] (zeroed here): rows per class [0..5],
 *                                stored elements of the heaviest NATURAL 35-row group [6] (the caller's "is it skewed" test)
 *   (caller)                     stable sort of rows by keys (spamd_sort_kv: longest rows first), class counts read back
 *   spamd_spmm_tiled_map_groups  groups of the layout for these class counts (host arithmetic)
 *   spamd_spmm_tiled_map_build   rowmap[groups * rows_per_group + 16] (-1 = unused slot), gload[groups + 1] (stored elements
 *                                per group, then 0): the caller's exclusive scan of gload is `vstart`
 *   spamd_spmm_tiled_inspect_mapped / spamd_spmm_tiled_mapped: the one-pass inspector and the executor on that layout. */
int spamd_spmm_tiled_map_stats(int idx_dtype, int64_t M, int64_t K, int64_t cap, const void* a_indptr, int64_t* keys,
                               int* rows, int64_t* stats, void* stream);
int64_t spamd_spmm_tiled_map_groups(const int64_t* class_counts /* host, 6 entries */);
int spamd_spmm_tiled_map_build(int idx_dtype, int64_t M, int64_t cap, const void* a_indptr, const int* rows_sorted,
                               const int64_t* class_counts /* host */, int* rowmap, int64_t* gload, void* stream);
int spamd_spmm_tiled_inspect_mapped(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t groups, const void* a_data,
                                    const void* a_indices, const void* a_indptr, const int* rowmap, const int64_t* vstart,
                                    void* state, int* blk_off, int* blocks, void* stream);
int spamd_spmm_tiled_mapped(int val_dtype, int64_t M, int64_t groups, int64_t K, int64_t N, const int* blocks,
                            const int* blk_off, const int* rowmap, const void* b, int64_t ldb, void* out, int64_t ldo,
                            unsigned flags, void* stream);
int spamd_spmm_tiled_keys(int64_t nnz, const int64_t* rowcol_keys, int64_t K, int64_t* tiled_keys, void* stream);
int spamd_spmm_tiled_lists(int val_dtype, int64_t nnz, const int64_t* tiled_keys_sorted, int64_t M, int64_t K,
                           int64_t* seg_start, int64_t* nblk, void* stream);
int spamd_spmm_tiled_pack(int val_dtype, int64_t nnz, const int64_t* tiled_keys_sorted, const void* vals_sorted,
                          const int64_t* seg_start, const int64_t* blk_off, int64_t total_blocks, int* blocks,
                          void* stream);
/* executor.  N = the PADDED width (whole panels; B provides that many columns, zero-padded).  flags: SPAMD_EXACT_MULADD,
 * SPAMD_TILED_GROUP_ENDS, SPAMD_TILED_INT32, bits 8..15 = lines of a list to prefetch (0: default), bits 16..23 = columns of the LAST panel
 * that are stored (0: all) - a narrower result is then written without padding: `out` holds N - panel + that many
 * columns per row (any count; until late round 4 float32 took even counts only). */
int spamd_spmm_tiled(int val_dtype, int64_t M, int64_t K, int64_t N, const int* blocks, const int* blk_off32,
                     const void* b, int64_t ldb, void* out, int64_t ldo, unsigned flags, void* stream);

/* ---------------------------------------------------------------------------------------
 * A10  NaN scan                    replaces `nan_check` (_common.py:51-69), the pass `matmul`
 *                                  runs over every operand before multiplying (:245-246).
 *   *flag (device int32) is set to 1 if any of the n values is NaN, else 0.  `data` must be
 *   16-byte aligned.  val_dtype: F32 | F64 (I32 | I64 accepted: flag = 0).
 * ------------------------------------------------------------------------------------- */
int spamd_has_nan(int val_dtype, int64_t n, const void* data, int* flag, void* stream);
/* The same scan, asynchronous: `host_flag` is the device-accessible address of a PINNED host int the caller has set
 * to 0; the kernel stores 1 there if it meets a NaN.  The caller records an event behind this call and reads the int
 * after waiting on that event alone, so the product queued behind the scan is not drained (reference: the blocking
 * `check_class_nan` pass of `matmul`, _common.py:245-246). */
int spamd_has_nan_async(int val_dtype, int64_t n, const void* data, int* host_flag, void* stream);

/* =======================================================================================
 * T1-T3 / A6  Canonicalisation and format conversion on 64-bit C-order linear keys.
 *   Replaces the reference's NumPy/numba integer pipeline: `linear_loc`
 *   (_coo/common.py:56-64), `COO._sort_indices/_sum_duplicates/_prune` (_coo/core.py:1294-1371),
 *   `_from_coo` (_compressed/compressed.py:25-77), `uncompress_dimension`/`_transpose`/
 *   `_convert_coords` (_compressed/convert.py:82-87,210-339), `COO.reshape/transpose`
 *   (_coo/core.py:725-807,1034-1111).  A sparse array is (keys[nnz] int64, data[nnz]);
 *   a transpose is a key permutation, a reshape is the identity on keys.
 *   All results are bit-exact integers.  `strides`/`dims`/`perm`/`axis_order` are HOST arrays
 *   of `ndim` (<= SPAMD_MAX_NDIM) entries, read during the call.
 * ===================================================================================== */

/* keys[p] = sum_d coords[axis_order[d]*coord_stride + p] * strides[d]   (coords: [ndim][nnz]) */
int spamd_coo_linearize(int idx_dtype, int ndim, int64_t nnz, const void* coords, int64_t coord_stride,
                        const int64_t* strides, const int32_t* axis_order, int64_t* keys, void* stream);
/* coords[d*coord_stride + p] = (keys[p] / strides[d]) % dims[d] */
int spamd_coo_delinearize(int idx_dtype, int ndim, int64_t nnz, const int64_t* keys, const int64_t* strides,
                          const int64_t* dims, void* coords, int64_t coord_stride, void* stream);
/* keys_out = C-order key after moving source axis perm[d] to destination position d */
int spamd_permute_keys(int ndim, int64_t nnz, const int64_t* keys_in, const int64_t* src_strides,
                       const int64_t* src_dims, const int32_t* perm, int64_t* keys_out, void* stream);
/* flag[0] = 1 if any coords[d][i] lies outside [0, dims[d]) (device int32[1]).  The reference constructor
 * (_coo/core.py:198-291) trusts its caller; this backend checks, because a bad coordinate becomes an
 * out-of-bounds key for spamd_scatter. */
int spamd_coords_check(int idx_dtype, int ndim, int64_t nnz, const void* coords, int64_t coord_stride,
                       const int64_t* dims, int* flag, void* stream);
/* flags2[0] = keys not non-decreasing, flags2[1] = some adjacent keys equal (device int32[2]) */
int spamd_keys_check(int64_t n, const int64_t* keys, int* flags2, void* stream);
/* flags[i] = 1 where a run of equal keys starts (int64 0/1, ready for spamd_exclusive_scan) */
int spamd_flag_heads(int64_t n, const int64_t* keys, int64_t* flags, void* stream);
/* flags[i] = 1 where data[i] is NOT bit-identical to the fill pattern (the reference's bit-wise
 * `equivalent`, _utils.py:448-452: -0.0 is not 0.0).  elem_bytes in {1,2,4,8,16}; fill_bits holds the element's low
 * 8 bytes, fill_bits_hi bytes 8..15 of a 16-byte element (complex128: the imaginary part) and is ignored otherwise. */
int spamd_flag_ne_bits(int elem_bytes, int64_t n, const void* data, uint64_t fill_bits, uint64_t fill_bits_hi,
                       int64_t* flags, void* stream);
/* *count (device int64) = number of elements bit-identical to the fill pattern: the cheap test that a prune has nothing
 * to do */
int spamd_count_eq_bits(int elem_bytes, int64_t n, const void* data, uint64_t fill_bits, uint64_t fill_bits_hi,
                        int64_t* count, void* stream);
/* dst[offsets[i]] = src[i] where flags[i] != 0  (stream compaction after an exclusive scan) */
int spamd_compact(int elem_bytes, int64_t n, const void* src, const int64_t* flags, const int64_t* offsets,
                  void* dst, void* stream);
/* dst[i] = src[perm[i]] ; dst[keys[i]] = src[i] */
int spamd_gather(int elem_bytes, int64_t n, const void* src, const int64_t* perm, void* dst, void* stream);
/* the same for every row of a [rows, n] matrix in one launch (the [ndim, nnz] coordinate matrix of a COO,
 * `coords[:, order]` / `coords[:, mask]` in the reference's `_sort_indices` / `_sum_duplicates`, _coo/core.py:1294-1353) */
int spamd_compact_rows(int elem_bytes, int rows, int64_t n, const void* src, int64_t ld_src, const int64_t* flags,
                       const int64_t* offsets, void* dst, int64_t ld_dst, void* stream);
int spamd_gather_rows(int elem_bytes, int rows, int64_t n, const void* src, int64_t ld_src, const int64_t* perm, void* dst,
                      int64_t ld_dst, void* stream);
int spamd_scatter(int elem_bytes, int64_t n, const void* src, const int64_t* keys, void* dst, void* stream);
/* sorted keys (row*C + col) -> indptr[R+1], indices[nnz]          (`_from_coo`, compressed.py:64-76) */
int spamd_keys_to_csr(int idx_dtype, int64_t nnz, const int64_t* keys, int64_t R, int64_t C, void* indptr,
                      void* indices, void* stream);
/* (indptr, indices) -> keys[p] = row(p)*C + indices[p]              (`uncompress_dimension`, convert.py:82-87) */
int spamd_csr_to_keys(int idx_dtype, int64_t R, int64_t nnz, const void* indptr, const void* indices, int64_t C,
                      int64_t* keys, void* stream);
/* sorted row ids -> int64 indptr[R+1]        (`cumsum(bincount(coords[0]))`, _common.py:452-458) */
int spamd_rows_to_indptr(int idx_dtype, int64_t nnz, const void* rows, int64_t R, int64_t* indptr, void* stream);
/* CSR <-> CSC of a 2-D compressed matrix with 4-byte values in one call: `GCXS.change_compressed_axes` / `_transpose`
 * (compressed.py:388-423, convert.py:210-273: uncompress, re-linearise, stable argsort, bincount + cumsum).  The input is
 * in (major, minor) order, so a stable sort on the minor index alone gives (minor, major) order.  data: nnz 4-byte values
 * (moved bit-wise), indices[nnz], indptr[n_major + 1] of idx_dtype; outputs of the same dtypes: out_data[nnz],
 * out_indices[nnz] (the major ids), out_indptr[n_minor + 1].  n_major, n_minor < 2^32; ws of spamd_csx_swap_ws_bytes(nnz). */
int64_t spamd_csx_swap_ws_bytes(int64_t nnz);
int spamd_csx_swap(int idx_dtype, int64_t n_major, int64_t n_minor, int64_t nnz, const void* data, const void* indices,
                   const void* indptr, void* out_data, void* out_indices, void* out_indptr, void* ws, int64_t ws_bytes,
                   void* stream);
/* the same for 8-byte values (float64 / int64: the reference's default value type): a 16-byte payload rides through the sort */
int64_t spamd_csx_swap8_ws_bytes(int64_t nnz);
int spamd_csx_swap8(int idx_dtype, int64_t n_major, int64_t n_minor, int64_t nnz, const void* data, const void* indices,
                   const void* indptr, void* out_data, void* out_indices, void* out_indptr, void* ws, int64_t ws_bytes,
                   void* stream);
/* Stable radix sort of (key, value) int64 pairs on key bits [0, end_bit); keys must be >= 0.
 * Replaces `np.argsort(linear, kind="mergesort")` (core.py:1315).  Workspace from *_ws_bytes. */
int64_t spamd_sort_pairs_ws_bytes(int64_t n);
int spamd_sort_pairs(int64_t n, const int64_t* keys_in, int64_t* keys_out, const int64_t* vals_in,
                     int64_t* vals_out, int end_bit, void* ws, int64_t ws_bytes, void* stream);
/* The same stable sort carrying a 4- or 8-byte VALUE as payload (used by SpGEMM's expand-sort-compress). */
int64_t spamd_sort_kv_ws_bytes(int val_bytes, int64_t n);
int spamd_sort_kv(int val_bytes, int64_t n, const int64_t* keys_in, int64_t* keys_out, const void* vals_in,
                  void* vals_out, int end_bit, void* ws, int64_t ws_bytes, void* stream);
int spamd_iota(int64_t n, int64_t* out, void* stream);
/* out[i] = in[0] + ... + in[i-1] for i in [0, n]; both arrays hold n+1 entries (in[n] ignored). */
int64_t spamd_scan_ws_bytes(int64_t n);
int spamd_exclusive_scan(int64_t n, const int64_t* in, int64_t* out, void* ws, int64_t ws_bytes, void* stream);
/* astype between F32/F64/I32/I64/U8 (C-cast, NumPy "unsafe"; to U8 = x != 0) */
int spamd_convert(int src_dtype, int dst_dtype, int64_t n, const void* src, void* dst, void* stream);

/* =======================================================================================
 * A7  Elementwise on canonical operands     replaces `_Elemwise` + `_match_arrays`
 *                                            (sparse/numba_backend/_umath.py:53-92,392-751)
 *   On sorted, duplicate-free linear keys the reference's mask enumeration / argsort / join /
 *   concatenate / re-sort collapses to one sorted-key UNION:
 *     out[k] = func(a[k] or fill_a, b[k] or fill_b), entries bit-equal to func(fill_a, fill_b)
 *     dropped afterwards (spamd_flag_ne_bits + spamd_compact).
 * ===================================================================================== */

/* pos[i] = number of h-keys < q[i]; match[i] = 1 iff q[i] occurs in h (h sorted + unique) */
int spamd_lower_bound_match(int64_t nq, const int64_t* q, int64_t nh, const int64_t* h, int64_t* pos,
                            int64_t* match, void* stream);
int spamd_invert_flags(int64_t n, const int64_t* in, int64_t* out, void* stream);
/* Output slot of every a / b element in the union and the union's keys.
 * posB/posA/matchB from spamd_lower_bound_match, ub = exclusive scan of (1 - matchB) (nb+1). */
int spamd_union_positions(int64_t na, const int64_t* ka, const int64_t* posB, int64_t nb, const int64_t* kb,
                          const int64_t* posA, const int64_t* matchB, const int64_t* ub, int64_t* slotA,
                          int64_t* slotB, int64_t* out_keys, void* stream);
int spamd_fill(int elem_bytes, int64_t n, void* out, uint64_t value_bits, void* stream);
/* out[i] = a[i] (op) b[i]; *_is_scalar broadcasts a 1-element device array.
 * op: 0 add 1 sub 2 mul 3 div 4 maximum 5 minimum 6 power 7 fmax 8 fmin 9 floor_divide 10 remainder 11 fmod (floats: NumPy's
 *     npy_divmod / npy_remainder statement for statement; integers: Python's floor rules, x // 0 = x % 0 = 0) 12 copysign
 *     13 hypot 14 arctan2 (12-14: F32 | F64 only) (out dtype = val_dtype);
 *     32 gt 33 ge 34 lt 35 le 36 eq 37 ne 38 logical_and 39 logical_or 40 logical_xor (out U8);
 *     64 bitwise_and 65 bitwise_or 66 bitwise_xor 67 left_shift 68 right_shift (integer dtypes; shift counts outside
 *     [0, bits) give 0 / the sign, as NumPy's).  No FMA contraction.  (reference: the ufunc itself, applied by
 *     `_elemwise_n_ary` sparse/numba_backend/_umath.py:420-470 to matched value arrays) */
int spamd_ewise_binary(int op, int val_dtype, int64_t n, const void* a, int a_is_scalar, const void* b,
                       int b_is_scalar, void* out, void* stream);
/* out[i] = f(a[i]); op: 0 negative 1 abs 2 sqrt 3 exp 4 expm1 5 log 6 log1p 7 sin 8 cos 9 tan 10 tanh
 *   11 sinh 12 cosh 13 arcsin 14 arctan 15 floor 16 ceil 17 rint 18 trunc 19 sign 20 square
 *   21 reciprocal 22 positive 23 log2 24 log10 25 exp2 26 arcsinh 27 arctanh 28 cbrt 29 deg2rad
 *   30 rad2deg (out dtype = val_dtype); 64 isnan 65 isinf 66 isfinite 67 logical_not 68 signbit (out U8) */
int spamd_ewise_unary(int op, int val_dtype, int64_t n, const void* a, void* out, void* stream);
/* out[i] = mask[i] ? a[i] : b[i] (`np.where` on aligned arrays of 1-, 4- or 8-byte elements, moved bit-wise; mask = 0/1 bytes;
 * *_is_scalar broadcasts a 1-element device array) */
int spamd_ewise_select(int elem_bytes, int64_t n, const void* mask_u8, const void* a, int a_is_scalar, const void* b,
                       int b_is_scalar, void* out, void* stream);

/* Merge-path form of the same union, fused with the function and the prune (the default path):
 *   nblocks = spamd_merge_num_blocks(na, nb);  part[nblocks+1] <- spamd_merge_partition;
 *   spamd_merge_union(fill=0, ...) -> counts[nblocks]; host: exclusive scan -> offsets, total;
 *   spamd_merge_union(fill=1, ...) -> out_keys[total] (strictly increasing), out_vals[total].
 *   out = func(a or fill_a, b or fill_b); entries bit-identical to fill_out are dropped.
 *   op codes as spamd_ewise_binary (6 = power and 9-14, 67, 68 are not available here); *_bits = raw bit patterns
 *   of the fill values in val_dtype (fill_out in the output dtype: U8 for ops 32..40). */
int64_t spamd_merge_num_blocks(int64_t na, int64_t nb);
int spamd_merge_partition(int64_t na, const int64_t* ka, int64_t nb, const int64_t* kb, int64_t* part,
                          void* stream);
int spamd_merge_union(int fill, int op, int val_dtype, int64_t na, const int64_t* ka, const void* va, int64_t nb,
                      const int64_t* kb, const void* vb, uint64_t fill_a_bits, uint64_t fill_b_bits,
                      uint64_t fill_out_bits, const int64_t* part, int64_t* counts, const int64_t* offsets,
                      int64_t* out_keys, void* out_vals, void* stream);
/* The same union in ONE launch and without a copy-back (csrc/merge.hip, MODE 3): every tile finds its own merge-path
 * diagonals, the tiles chain their output offsets by look-back, and the kernel leaves its workspace zeroed for the next
 * call.  ws = int64[3 + capacity], capacity >= spamd_merge_fused_blocks(na, nb), all zero before the first use; calls that
 * share a workspace must be ordered on one stream.  *total_dev (device) receives the number of outputs; total_host, if not
 * null, is pinned host memory mapped to the device that receives it too (system-scope release store) - the host may spin on
 * it.  out_keys / out_vals hold na + nb elements.  Replaces `_match_arrays` + the mask loop of `_Elemwise`,
 * _umath.py:53-92,576-654, for canonical same-shape operands. */
/* tiles the fused form below runs (its workspace holds 3 + that many int64) */
int64_t spamd_merge_fused_blocks(int64_t na, int64_t nb);
int spamd_merge_union_fused(int op, int val_dtype, int64_t na, const int64_t* ka, const void* va, int64_t nb,
                            const int64_t* kb, const void* vb, uint64_t fill_a_bits, uint64_t fill_b_bits,
                            uint64_t fill_out_bits, int64_t* ws, int64_t* total_dev, int64_t* total_host, int64_t* out_keys,
                            void* out_vals, void* stream);

/* ---------------------------------------------------------------------------------------
 * A8  Grouped reduce     replaces `_calc_counts_invidx` + `_grouped_reduce` / `ufunc.reduceat`
 *                        (sparse/numba_backend/_coo/core.py:1601-1661; compressed.py:354-386)
 *   heads[i] = 1 where a run starts, offsets = its exclusive scan, nseg = number of runs.
 *   out[g] = data[s_g] op data[s_g+1] op ...; counts[g] = run length (may be NULL).
 *   Short runs: one thread per run, strictly left to right (bit-identical to reduceat).
 *   Long runs (n/nseg >= 24 and seg_start_ws != NULL, nseg+1 int64): one wave per run (tree order).
 *   op: 0 add 1 multiply 2 maximum 3 minimum 4 logical_or 5 logical_and 6 fmax 7 fmin (NaN-skipping).
 * ------------------------------------------------------------------------------------- */
int spamd_segment_reduce(int op, int val_dtype, int64_t n, const void* data, const int64_t* heads,
                         const int64_t* offsets, int64_t nseg, void* out, int64_t* counts,
                         int64_t* seg_start_ws, void* stream);

/* A8 epilogue in one launch: fold the implicit fill entries of each group into vals[i] (reference
 * _sparse_array.py:405-422).  counts[i] = stored elements of group i, n_cols = elements per group; the fill value
 * is passed both as double and as int64 (the one matching val_dtype is used).  add/multiply use the closed form
 * in the work dtype (float64 for floating results), the other ops fold fv in once where counts[i] != n_cols. */
int spamd_reduce_fill(int op, int val_dtype, int64_t n, void* vals, const int64_t* counts, int64_t n_cols, double fill_f,
                      int64_t fill_i, void* stream);
/* The same fold-in with the number of groups still on the device (*n_dev, as spamd_group_reduce leaves it; the grid is
 * sized for n_max), plus *n_eq (device int64, zeroed here) = how many folded results are bit-identical to
 * result_fill_bits (n_eq == n_dev + 1, the second word of spamd_group_reduce's n_groups, is already zero and not cleared
 * again): the host reads both numbers in one copy, and the result container's prune (`COO(..., prune=True)`,
 * _coo/core.py:705-716) has nothing to do when *n_eq == 0. */
int spamd_reduce_fill_count(int op, int val_dtype, int64_t n_max, const int64_t* n_dev, void* vals, const int64_t* counts,
                            int64_t n_cols, double fill_f, int64_t fill_i, uint64_t result_fill_bits, int64_t* n_eq,
                            void* stream);
/* A8 in one pass: runs of equal (keys[i] / divisor) over SORTED keys are reduced together with their lengths
 * (two streaming passes, csrc/group_reduce.hip; fp sums in a fixed, reproducible order).  Outputs hold up to n entries;
 * n_groups = device int64[2]: [0] receives the number of runs, [1] is cleared (the counter spamd_reduce_fill_count then
 * accumulates into, so that the caller reads both numbers with one copy).  op as for spamd_segment_reduce;
 * val_dtype F32 | F64 | I32 | I64 | U8.  keys < key_bound (0 = unknown; below 2^53 the ids are computed in
 * double precision).  keys and data 16-byte aligned.  Workspace from spamd_group_reduce_ws_bytes. */
int64_t spamd_group_reduce_ws_bytes(int val_dtype, int64_t n);
int spamd_group_reduce(int op, int val_dtype, int64_t n, const int64_t* keys, int64_t divisor, int64_t key_bound,
                       const void* data, int64_t* group_ids, void* values, int64_t* counts, int64_t* n_groups, void* ws,
                       int64_t ws_bytes, void* stream);
/* The same reduction when EVERY axis is reduced (`x.sum()`, `x.max()`, ... with axis=None; reference _sparse_array.py:372-437
 * with axis = all axes: one group): the keys are not read.  Outputs in spamd_group_reduce's form, one entry each:
 * group_ids[0] = 0, values[0], counts[0] = n, n_groups = {1, 0} ({0, ...} for n == 0).  One launch; pieces of the values
 * are folded in a fixed order (reproducible).  ws: spamd_reduce_all_ws_bytes() bytes, 16-byte aligned, ZEROED once by the
 * caller and then reusable by successive calls on one stream (the kernel leaves its ticket word zero). */
int64_t spamd_reduce_all_ws_bytes(void);
int spamd_reduce_all(int op, int val_dtype, int64_t n, const void* data, int64_t* group_ids, void* values, int64_t* counts,
                     int64_t* n_groups, void* ws, int64_t ws_bytes, void* stream);

/* n (1..16) device int64 words to the host WITHOUT a blocking copy: queued behind the stream's work, one thread stores
 * dev_words[0 .. n-1] into host_words[0 .. n-1] and then, with release semantics, `marker` into host_words[n].
 * host_words = n + 1 int64 of PINNED host memory mapped to the device (hipHostMalloc / torch pin_memory); the caller picks a
 * marker host_words[n] does not hold (a per-call sequence number) and spins until it appears.  The read-backs of
 * data-dependent sizes (`SparseArray.reduce`'s group count, _coo/core.py:693-723) use it: a `.item()` costs a stream
 * synchronisation plus a copy command, ~20 us - as much as the kernels of a 10^6-element reduction. */
int spamd_deliver_words(const int64_t* dev_words, int n, int64_t* host_words, int64_t marker, void* stream);

/* out = in^T for a dense row-major matrix (elements of 1 / 2 / 4 / 8 bytes, moved bit-wise; ld_in >= cols, ld_out >= rows):
 * out[c * ld_out + r] = in[r * ld_in + c].  `dense @ sparse` runs as (sparse^T @ dense^T)^T (the reference's dispatch,
 * sparse/numba_backend/_common.py:339-503, hands the dense operand to the kernels transposed and contiguous,
 * `np.ascontiguousarray`, _common.py:744); this is that copy through 64 x 64 LDS tiles. */
int spamd_transpose_2d(int elem_bytes, int64_t rows, int64_t cols, const void* in, int64_t ld_in, void* out, int64_t ld_out,
                       void* stream);

/* Hub rows (round 6, csrc/hot_rows.hip): `_dot.py` multiplies an operand with a few rows far longer than the rest in two
 * parts - the matrix without them and the hot rows cut into pieces that are rows of their own - through the CSR x dense kernels
 * above (reference `_dot_csr_ndarray`, _common.py:720-755: one core walks such a row as one wave does here); this adds the
 * pieces' results into the hot rows of the result: out[rows[h], :] = sum over v in [vfirst[h], vfirst[h + 1]) of part[v, :],
 * in piece order.  val_dtype F32 | F64 | I32 | I64; vfirst has n_hot + 1 entries. */
int spamd_hot_rows_combine(int val_dtype, int64_t n_hot, int64_t n_cols, const void* part, int64_t ld_part,
                           const int64_t* vfirst, const int64_t* rows, void* out, int64_t ld_out, void* stream);

/* ---------------------------------------------------------------------------------------
 * A4 / A5  sparse x sparse     replaces `_csr_csr_count_nnz` + `_dot_csr_csr` / `_dot_coo_coo`
 *                              (sparse/numba_backend/_common.py:543-570,639-717,907-976)
 *   Expand-sort-compress: the two entry points below produce, for the A elements [p0, p0+np),
 *   (1) cnt[i] = nnz of B row a_indices[p0+i]; after the caller's exclusive scan (offsets, P):
 *   (2) keys[t] = a_rows[p]*n_col + b_col, vals[t] = a*b for every product t in [0, P).
 *   A stable sort by key + spamd_segment_reduce(add) then gives every output element summed in
 *   the reference's order (bit-identical), with rows sorted by column.
 * ------------------------------------------------------------------------------------- */
int spamd_spgemm_count(int idx_dtype, int64_t p0, int64_t np, const void* a_indices, const void* b_indptr,
                       int64_t* cnt, void* stream);
int spamd_spgemm_expand(int val_dtype, int idx_dtype, int64_t p0, int64_t np, const void* a_data,
                        const void* a_indices, const int64_t* a_rows, const void* b_data, const void* b_indices,
                        const void* b_indptr, const int64_t* offsets, int64_t P, int64_t n_col, int64_t* keys,
                        void* vals, void* stream);

/* A4 / A5, row-local form (csrc/spgemm_rows.hip): the products of an output row are expanded, ordered by column
 * (bucket by the high column bits, then rank inside the small buckets: hand-written, no library sort) and summed
 * inside LDS by one workgroup; bit-identical to the expand-sort-compress above, without writing or sorting the
 * products in HBM.  Rows with more products than spamd_spgemm_rows_capacity(...) are skipped and rows with a column that
 * collects more than 64 products are declined (nnz_row = -1): the caller computes those with the global form and merges
 * them with spamd_spgemm_unpack.  n_col < 2^31 - 1.
 *   spamd_spgemm_row_products: prod[n_row + 1] (last entry zeroed) and maxes[2] = {max prod, longest A row};
 *   caller: prod_off = spamd_exclusive_scan(prod), scratch tmp_cols/tmp_vals of prod_off[n_row] entries;
 *   spamd_spgemm_rows: rows of C into the scratch at prod_off[row], their lengths into nnz_row[n_row + 1];
 *   caller: out_indptr = spamd_exclusive_scan(nnz_row);
 *   spamd_spgemm_pack: scratch -> (out_indices int64, out_data), columns ascending inside every row; zero_count
 *     (optional, one device word, zeroed here) receives the number of all-zero-bits values written: what the
 *     `GCXS(..., prune=True)` of reference _common.py:374-379 then need not count in a pass of its own. */
int spamd_spgemm_row_products(int idx_dtype, int64_t n_row, const void* a_indptr, const void* a_indices,
                              const void* b_indptr, int64_t* prod, int64_t* maxes, void* stream);
int64_t spamd_spgemm_rows_capacity(int val_dtype, int64_t n_col, int64_t max_arow);
int spamd_spgemm_rows(int val_dtype, int idx_dtype, int64_t n_row, int64_t n_col, const void* a_indptr,
                      const void* a_indices, const void* a_data, const void* b_indptr, const void* b_indices,
                      const void* b_data, const int64_t* prod_off, int64_t max_prod, int64_t max_arow, int* tmp_cols,
                      void* tmp_vals, int64_t* nnz_row, void* stream);
/* which rows are left to the global form: flags[n_row + 1] for spamd_exclusive_scan + spamd_compact, counts[3] = {rows,
 * their products, rows heavy only by their A length}; nnz_row NULL before the row kernel, else rows with -1 count too */
int spamd_spgemm_classify_rows(int idx_dtype, int64_t n_row, const int64_t* prod, const void* a_indptr,
                               const int64_t* nnz_row, int64_t cap, int64_t* flags, int64_t* counts, void* stream);
/* rows too heavy for the row-local kernel: computed by the global form, given as CSR over all n_row rows (int64
 * columns), copied into the scratch at their product offset; nnz_row[row] is set for the listed rows */
int spamd_spgemm_unpack(int val_dtype, int64_t n_heavy, const int64_t* heavy_rows, const int64_t* src_indptr,
                        const int64_t* src_indices, const void* src_data, const int64_t* prod_off, int* tmp_cols,
                        void* tmp_vals, int64_t* nnz_row, void* stream);
int spamd_spgemm_pack(int val_dtype, int64_t n_row, const int64_t* prod_off, const int64_t* out_indptr,
                      const int* tmp_cols, const void* tmp_vals, int64_t* out_indices, void* out_data,
                      int64_t* zero_count, void* stream);

/* A4 / A5, row-local form for wide rows of a matrix with at most 2^20 columns (csrc/spgemm_bitmap.hip; replaces the same
 * reference functions, sparse/numba_backend/_common.py:543-570,639-717).  A persistent workgroup per CU keeps a bitmap
 * of the output row's columns in LDS: a column's position in the sorted row is a popcount, the row's length is known
 * before anything is written, rows take tickets and find their offset by a look-back over one state word per row, and
 * every row is written ONCE, in place - no scratch rows, no pack, no scan of row lengths.  Values are summed in the order
 * of A's elements (bit-identical to the other two forms).
 *   spamd_spgemm_bitmap_limits(val_dtype, which): 0 = products per row, 1 = A elements per row, 2 = columns,
 *     3 = products per row that may share an output element with an earlier product (checked inside the kernel),
 *     4 / 5 / 6 = products, columns and such products per PART of a row in the split form (below);
 *   spamd_spgemm_bitmap: out_indptr[n_row + 1]; out_indices (int64) / out_data with room for EVERY product (the caller
 *     trims to out_indptr[n_row]); work = n_row + 32 int64 words, zeroed here: afterwards work[1] != 0 = a row was outside
 *     the limits (discard the result, use spamd_spgemm_rows), work[2] = values written whose bits are all zero. */
int64_t spamd_spgemm_bitmap_limits(int val_dtype, int which);
/* A4 / A5 for SMALL operands (round 5): the reference's row loop (`_dot_csr_csr`, _common.py:639-717: a dense accumulator per
 * output row, the row's A elements in storage order) with a WAVE per output row, accumulator and touched-column bitmap in LDS,
 * rows written column-sorted to their final place through a look-back over the rows - ONE launch, one read-back.  The sizes
 * of the reference's own benchmark (benchmarks/test_benchmark_coo.py:9-40) are launch-bound on the general path.
 * out_indices / out_data: room for n_row * n_col entries; work: n_row + 4 words (zeroed here); afterwards work[1] != 0 = failed
 * (a B row not strictly ascending or out of range: discard, use the general path), work[2] = exact zeros written.
 * n_col <= spamd_spgemm_small_max_cols(val_dtype). */
int64_t spamd_spgemm_small_max_cols(int val_dtype);
int spamd_spgemm_small(int val_dtype, int idx_dtype, int64_t n_row, int64_t n_col, const void* a_indptr, const void* a_indices,
                       const void* a_data, const void* b_indptr, const void* b_indices, const void* b_data, int64_t* work,
                       int64_t* out_indptr, int64_t* out_indices, void* out_data, void* stream);

/* parts = 1: whole rows, one 1024-thread workgroup per CU (n_col <= limit 2).  parts > 1: every row in `parts`
 * column ranges (ceil(n_col / parts) rounded up to 256 <= limit 5), 512-thread workgroups, two per CU; bsplit = n_inner *
 * (parts - 1) words of the index type (workspace, filled here: where each B row crosses a range boundary), n_inner = rows
 * of B; limits 4 / 6 are per part.  work = n_row * parts + 32 int64 words. */
int spamd_spgemm_bitmap(int val_dtype, int idx_dtype, int64_t n_row, int64_t n_inner, int64_t n_col, int parts,
                        const void* a_indptr, const void* a_indices, const void* a_data, const void* b_indptr,
                        const void* b_indices, const void* b_data, void* bsplit, int64_t* work, int64_t* out_indptr,
                        int64_t* out_indices, void* out_data, void* stream);

/* A8 / A6 (round 5): keys of a canonical COO with its LEADING axes moved last, sorted, without a sort.  The reference's
 * reduction over axes 0 .. m-1 transposes the kept axes first and sorts every stored element (`COO.reduce` ->
 * `_reduce_calc`, _coo/core.py:693-723; `transpose` + `reshape`, :1601-1661).  Sorted keys = S sorted runs (one per index of the
 * leading axes), merged by ranges of kept cells in LDS (csrc/lead_rotate.hip).  keys[n] sorted and duplicate-free,
 * key = s * P + c; out_keys[n] = c * S + s ascending, out_vals in that order (val_bytes 4 / 8, bit-wise).
 * cells_per_range: a power of two <= spamd_keys_lead_last_limits(1); bounds: S * (ceil(P / cells_per_range) + 1) ints of
 * workspace; S <= limits(0); n < 2^31.  *failed (device, zeroed here) != 0: a range held more than limits(2) elements or a
 * cell more than 64 - out_* are incomplete, use spamd_permute_keys + spamd_sort_kv. */
int64_t spamd_keys_lead_last_limits(int which);
int spamd_keys_lead_last(int val_bytes, int64_t n, const int64_t* keys, const void* vals, int64_t S, int64_t P,
                         int64_t cells_per_range, int* bounds, int64_t* out_keys, void* out_vals, int64_t* failed, void* stream);

/* N3 / A7 (round 5): a canonical COO broadcast to a larger shape in one pass, sorted by construction (csrc/broadcast.hip).
 * Replaces the operand expansion of the reference's elementwise broadcasting (`broadcast_to`, _umath.py:344-389;
 * `_get_expanded_coords_data`, :96-167).  The target's axes, left to right, are [B0][K1][B1][K2][B2] - B: groups of broadcast
 * axes, K: groups of the operand's own axes, sizes b0 k1 b1 k2 b2 (1 for an empty group); keys[n] sorted and duplicate-free
 * over (K1, K2); out_keys[n b0 b1 b2] over the target's axes, ascending; out_vals the replicated values (val_bytes 1/2/4/8). */
int spamd_coo_broadcast(int val_bytes, int64_t n, const int64_t* keys, const void* vals, int64_t b0, int64_t k1, int64_t b1,
                        int64_t k2, int64_t b2, int64_t* out_keys, void* out_vals, void* stream);

/* N1 (round 5): a dense array's stored elements in one pass (reference `COO.from_numpy`, _coo/core.py:341-384: the elements not
 * equal to the fill value and their positions).  vals[n] (val_bytes 1 / 2 / 4 / 8); fill_bits: the fill value's bit pattern;
 * cmp_mask: the bits compared (all ones = bit-identity; a floating type's mask without the sign bit + fill 0 = `value != 0`, the
 * test of the reference's COO-returning products `_common.py:1049, 1139`);
 * out_keys / out_vals: room for n entries, the first work[1] are written (indices ascending); work:
 * spamd_dense_nonfill_work_words(n) int64 words, zeroed here (a ticket, the count, one look-back word per 2048 elements). */
int64_t spamd_dense_nonfill_work_words(int64_t n);
int spamd_dense_nonfill(int val_bytes, int64_t n, const void* vals, uint64_t fill_bits, uint64_t cmp_mask, int64_t* work,
                        int64_t* out_keys, void* out_vals, void* stream);

/* ---------------------------------------------------------------------------------------
 * A9  SDDMM      out[n] = s[n] * sum_k A[rows[n], k] * Bt[cols[n], k]
 *   replaces the reference's formulation `s * (a @ b)` (examples/sddmm_example.py:51-52: a dense
 *   BLAS GEMM of the full M x N product + `_Elemwise` gather, _umath.py:602-633).
 *   A is M x K row-major (lda), Bt = B^T is N x K row-major (ldb): both K-contiguous, 16-byte
 *   aligned rows.  in_dtype BF16|F32 -> fp32 accumulate and F32 mask/out; F64 -> F64.
 * ------------------------------------------------------------------------------------- */
int spamd_sddmm(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows, const void* cols,
                const void* s_data, const void* A, int64_t lda, const void* Bt, int64_t ldb, int64_t K, void* out,
                void* stream);

/* A9 in column-panel order (same result, element for element, as spamd_sddmm; same reference formulation,
 * examples/sddmm_example.py:51-52).  When Bt does not fit an XCD's L2, a mask walked in row-major order fetches every
 * Bt row from the Infinity Cache.  spamd_sddmm_panel_keys gives keys[n] = cols[n] / width; the caller sorts them
 * stably (spamd_sort_pairs) into `perm`, gathers rows/cols/values by it (rows_p, cols_p, s_p), and spamd_sddmm_panels
 * walks the mask one panel of `width` Bt rows at a time, lane groups taking `chunk` elements per turn (<= 0: default):
 * out[perm[n]] = s_p[n] * <A[rows_p[n]], Bt[cols_p[n]]>, i.e. out keeps the mask's own order.  perm/rows_p/cols_p
 * depend on the coordinates only.  SPAMD_EINVAL when K has no row-cached kernel (call spamd_sddmm).
 * XCD-private panels (optional): with per_xcd = ceil(panels / 8) the keys are XCD-major, (panel % 8) * per_xcd + panel / 8,
 * xcd_first = int64[9] (device) holds the first element of each XCD's range in the sorted order (and the total) and
 * xcd_max the longest range; workgroup b then takes piece b / 8 of the range of XCD b % 8 (the observed placement: for
 * speed only), so that a panel's Bt rows are fetched into one L2 instead of eight.  NULL = all XCDs share every panel. */
int spamd_sddmm_has_panels(int in_dtype, int64_t K); /* 1: spamd_sddmm_panels has a kernel for this K */
int spamd_sddmm_panel_keys(int idx_dtype, int64_t nnz, const void* cols, int64_t width, int64_t per_xcd, void* keys,
                           void* stream);
/* Rows of exactly 1 KB (fp32 K = 256, fp64 K = 128, bf16 K = 512): with `part` (nnz words of the accumulator type - fp32, or
 * fp64 for F64 operands - of caller-owned scratch) the product runs as TWO passes over 512-byte half-rows (the first leaves its
 * sums in `part` in panel order, the second adds its half and writes s * sum), so that a panel holds twice the Bt rows in the
 * same L2 bytes and A is streamed half as often (config 4 fp32: 6.4 GB -> ~2.5 GB of fabric traffic); the panels are then
 * built for the row length spamd_sddmm_panel_row_bytes() reports (512).  part = NULL, or any other row length: one pass. */
int64_t spamd_sddmm_panel_row_bytes(int in_dtype, int64_t K); /* bytes of a Bt row the panel width is computed for */
int spamd_sddmm_panels(int in_dtype, int s_dtype, int idx_dtype, int64_t nnz, const void* rows_p, const void* cols_p,
                       const int64_t* perm, const void* s_p, const void* A, int64_t lda, const void* Bt, int64_t ldb,
                       int64_t K, int64_t chunk, const int64_t* xcd_first, int64_t xcd_max, void* part, void* out,
                       void* stream);

/* A9, dense-tile form (north_star: "MFMA used only on the dense tile of SDDMM"): the 32 x 32 tiles of the mask that
 * hold at least `threshold` samples are computed as one 32 x 32 x K bf16 product on the matrix cores
 * (v_mfma_f32_32x32x16_bf16, fp32 accumulate) and sampled from LDS; every other sample goes to spamd_sddmm.
 * Same reference formulation (examples/sddmm_example.py:51-52).  Plan: spamd_sddmm_tile_keys -> stable sort of the keys
 * with the sample index as payload (spamd_sort_pairs) -> spamd_flag_heads / spamd_exclusive_scan / spamd_compact give
 * seg_start[nseg + 1] -> spamd_sddmm_tile_classify -> scan + compact give the list of dense tiles and of left-over
 * samples.  bf16 operands only, K a multiple of 16, 16-byte aligned rows. */
int spamd_sddmm_tile_size(void);
int spamd_sddmm_tile_keys(int idx_dtype, int64_t nnz, const void* rows, const void* cols, int64_t tile_cols,
                          int64_t* keys, void* stream);
int spamd_sddmm_tile_classify(int64_t nseg, const int64_t* seg_start, int64_t threshold, int64_t* tile_flag,
                              int64_t* sample_flag, void* stream);
int spamd_sddmm_mfma_tiles(int idx_dtype, int64_t ntiles, const int64_t* tiles, const int64_t* seg_start,
                           const int64_t* keys_sorted, const int64_t* perm, int64_t tile_cols, int64_t M, int64_t N,
                           const void* rows, const void* cols, const float* s_data, const void* A, int64_t lda,
                           const void* Bt, int64_t ldb, int64_t K, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSE_AMD_H */
