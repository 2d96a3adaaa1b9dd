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

/* error codes (negative; positive values are hipError_t) */
#define SPAMD_EINVAL (-1)  /* bad size / null pointer / misaligned */
#define SPAMD_ETYPE (-2)   /* unsupported dtype combination */
#define SPAMD_EWS (-3)     /* workspace too small */

/* flags */
#define SPAMD_EXACT_MULADD 1u /* separate IEEE mul + add (bit-exact vs the reference's
                                 non-contracted loop) instead of fused multiply-add */

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
 *   val_dtype: F32 | F64 | I32 | I64.   idx_dtype: I32 | I64 (a_indices and a_indptr).
 * ------------------------------------------------------------------------------------- */
int spamd_spmm_csr(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N,
                   const void* a_data, const void* a_indices, const void* a_indptr,
                   const void* b, int64_t ldb, void* out, int64_t ldo,
                   unsigned flags, void* stream);

/* ---------------------------------------------------------------------------------------
 * A10  NaN scan                    replaces `nan_check` (_common.py:51-69), the pass `matmul`
 *                                  runs over every operand before multiplying (:245-246).
 *   *flag (device int32) is set to 1 if any of the n values is NaN, else 0.  `data` must be
 *   16-byte aligned.  val_dtype: F32 | F64 (I32 | I64 accepted: flag = 0).
 * ------------------------------------------------------------------------------------- */
int spamd_has_nan(int val_dtype, int64_t n, const void* data, int* flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSE_AMD_H */
