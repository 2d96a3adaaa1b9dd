/*
 * oracle.c — CPU restatement of the pydata/sparse numba_backend dot kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the `cpu_baseline`
 * ("port") leg of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (sparse_amd/) never does.
 *
 * Each function restates — loop nest for loop nest, same iteration order, same accumulator
 * dtype, separate multiply and add (build with -ffp-contract=off) — the reference kernel
 * it cites, so its floating-point results are bit-identical to the reference run under the
 * interpreter (pinned by tests/test_oracle_pin.py against tests/golden/ fixtures that
 * oracle/gen_golden.py produced by executing the real reference source).
 * Single-threaded, like the reference (`@numba.jit(nopython=True, nogil=True)`, no prange).
 *
 * Build: see oracle/build.py — the checker is `gcc -O3 -ffp-contract=off -shared -fPIC` (generic x86-64, so the .so
 * built in the container also runs on the GPU box); bench.py's cpu_baseline leg rebuilds it ON the timed box with
 * -march=native added (oracle/_build/liboracle_native.so).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define CAT3(a, b, c) CAT(CAT(a, b), c)
#define CAT5(a, b, c, d, e) CAT(CAT3(a, b, c), CAT(d, e))

typedef float f32;
typedef double f64;
typedef int32_t i32;
typedef int64_t i64;

/* ------------------------------------------------------------------------------------
 * Generic bodies, instantiated per (value type T, index type I) below.
 * ---------------------------------------------------------------------------------- */

/* _dot_csr_ndarray — reference sparse/numba_backend/_common.py:744-753
 *   out = zeros(out_shape); for i: for k in indptr[i]..indptr[i+1]: for j: out[i,j] += v*b[ind,j] */
#define DEF_CSR_NDARRAY(T, I)                                                                   \
  void CAT5(oracle_dot_csr_ndarray_, T, _, I, )(i64 M, i64 N, const T* a_data,                  \
                                                const I* a_indices, const I* a_indptr,          \
                                                const T* b, i64 ldb, T* out) {                  \
    memset(out, 0, (size_t)M * (size_t)N * sizeof(T));                                          \
    for (i64 i = 0; i < M; ++i) {                                                               \
      T* val = out + i * N;                                                                     \
      for (i64 k = a_indptr[i]; k < a_indptr[i + 1]; ++k) {                                     \
        const T* brow = b + (i64)a_indices[k] * ldb;                                            \
        const T v = a_data[k];                                                                  \
        for (i64 j = 0; j < N; ++j) val[j] += v * brow[j];                                      \
      }                                                                                         \
    }                                                                                           \
  }

/* _dot_csc_ndarray — reference _common.py:893-902
 *   A is (M x K) stored by columns: a_indptr has K+1 entries, a_indices are ROW ids.
 *   for i in range(K): for k in col i: out[ind,:] += v * b[i,:] */
#define DEF_CSC_NDARRAY(T, I)                                                                   \
  void CAT5(oracle_dot_csc_ndarray_, T, _, I, )(i64 M, i64 K, i64 N, const T* a_data,           \
                                                const I* a_indices, const I* a_indptr,          \
                                                const T* b, i64 ldb, T* out) {                  \
    memset(out, 0, (size_t)M * (size_t)N * sizeof(T));                                          \
    for (i64 i = 0; i < K; ++i) {                                                               \
      const T* brow = b + i * ldb;                                                              \
      for (i64 k = a_indptr[i]; k < a_indptr[i + 1]; ++k) {                                     \
        T* val = out + (i64)a_indices[k] * N;                                                   \
        const T v = a_data[k];                                                                  \
        for (i64 j = 0; j < N; ++j) val[j] += v * brow[j];                                      \
      }                                                                                         \
    }                                                                                           \
  }

/* _dot_coo_ndarray — reference _common.py:999-1012.
 *   coords is [2, nnz] (rows sorted).  The reference receives array2 = B.T (N x K strided
 *   view) and reads array2[oidx2, coords[1,k]] == B[coords[1,k], oidx2]; here B is passed
 *   directly as K x N row-major with leading dimension ldb.  For every run of equal row ids,
 *   for every output column, walk the run in storage order. */
#define DEF_COO_NDARRAY(T, I)                                                                   \
  void CAT5(oracle_dot_coo_ndarray_, T, _, I, )(i64 nnz, const I* rows, const I* cols,          \
                                                const T* data, const T* b, i64 ldb, i64 M,      \
                                                i64 N, T* out) {                                \
    memset(out, 0, (size_t)M * (size_t)N * sizeof(T));                                          \
    i64 didx1 = 0;                                                                              \
    while (didx1 < nnz) {                                                                       \
      const I oidx1 = rows[didx1];                                                              \
      const i64 didx1_curr = didx1;                                                             \
      for (i64 oidx2 = 0; oidx2 < N; ++oidx2) {                                                 \
        didx1 = didx1_curr;                                                                     \
        while (didx1 < nnz && rows[didx1] == oidx1) {                                           \
          out[(i64)oidx1 * N + oidx2] += data[didx1] * b[(i64)cols[didx1] * ldb + oidx2];       \
          ++didx1;                                                                              \
        }                                                                                       \
      }                                                                                         \
      if (N == 0) { /* degenerate: skip the run */                                              \
        while (didx1 < nnz && rows[didx1] == oidx1) ++didx1;                                    \
      }                                                                                         \
    }                                                                                           \
  }

/* _csr_csr_count_nnz — reference _common.py:559-570 (symbolic SpGEMM pass).
 * Also fills row_nnz[i] (the reference only returns the total). */
#define DEF_CSR_CSR_COUNT(I)                                                                    \
  i64 CAT(oracle_csr_csr_count_nnz_, I)(i64 n_row, i64 n_col, const I* a_indices,               \
                                        const I* b_indices, const I* a_indptr,                  \
                                        const I* b_indptr, i64* row_nnz) {                      \
    i64 nnz = 0;                                                                                \
    i64* mask = (i64*)malloc(sizeof(i64) * (size_t)(n_col > 0 ? n_col : 1));                    \
    for (i64 k = 0; k < n_col; ++k) mask[k] = -1;                                               \
    for (i64 i = 0; i < n_row; ++i) {                                                           \
      i64 rn = 0;                                                                               \
      for (i64 p = a_indptr[i]; p < a_indptr[i + 1]; ++p) {                                     \
        const i64 j = a_indices[p];                                                             \
        for (i64 q = b_indptr[j]; q < b_indptr[j + 1]; ++q) {                                   \
          const i64 k = b_indices[q];                                                           \
          if (mask[k] != i) {                                                                   \
            mask[k] = i;                                                                        \
            ++rn;                                                                               \
          }                                                                                     \
        }                                                                                       \
      }                                                                                         \
      if (row_nnz) row_nnz[i] = rn;                                                             \
      nnz += rn;                                                                                \
    }                                                                                           \
    free(mask);                                                                                 \
    return nnz;                                                                                 \
  }

/* _dot_csr_csr — reference _common.py:666-715 (numeric Gustavson pass with the intrusive
 * linked list; rows come out in reverse discovery order; explicit zeros are kept here and
 * pruned later by the GCXS constructor; the fully-dense result has each row reversed,
 * :709-714).  out_indices/out_indptr are intp (int64) as in the reference (:669-671).
 * Caller allocates out_data/out_indices with the count from oracle_csr_csr_count_nnz. */
#define DEF_CSR_CSR(T, I)                                                                       \
  i64 CAT5(oracle_dot_csr_csr_, T, _, I, )(i64 n_row, i64 n_col, const T* a_data,               \
                                           const T* b_data, const I* a_indices,                 \
                                           const I* b_indices, const I* a_indptr,               \
                                           const I* b_indptr, T* out_data, i64* out_indices,    \
                                           i64* out_indptr, i64 nnz_alloc) {                    \
    i64* next_ = (i64*)malloc(sizeof(i64) * (size_t)(n_col > 0 ? n_col : 1));                   \
    T* sums = (T*)calloc((size_t)(n_col > 0 ? n_col : 1), sizeof(T));                           \
    for (i64 k = 0; k < n_col; ++k) next_[k] = -1;                                              \
    i64 nnz = 0;                                                                                \
    out_indptr[0] = 0;                                                                          \
    for (i64 i = 0; i < n_row; ++i) {                                                           \
      i64 head = -2, length = 0;                                                                \
      for (i64 p = a_indptr[i]; p < a_indptr[i + 1]; ++p) {                                     \
        const i64 j = a_indices[p];                                                             \
        const T av = a_data[p];                                                                 \
        for (i64 q = b_indptr[j]; q < b_indptr[j + 1]; ++q) {                                   \
          const i64 k = b_indices[q];                                                           \
          sums[k] += av * b_data[q];                                                            \
          if (next_[k] == -1) {                                                                 \
            next_[k] = head;                                                                    \
            head = k;                                                                           \
            ++length;                                                                           \
          }                                                                                     \
        }                                                                                       \
      }                                                                                         \
      for (i64 t = 0; t < length; ++t) {                                                        \
        if (next_[head] != -1) {                                                                \
          out_indices[nnz] = head;                                                              \
          out_data[nnz] = sums[head];                                                           \
          ++nnz;                                                                                \
        }                                                                                       \
        const i64 temp = head;                                                                  \
        head = next_[head];                                                                     \
        next_[temp] = -1;                                                                       \
        sums[temp] = 0;                                                                         \
      }                                                                                         \
      out_indptr[i + 1] = nnz;                                                                  \
    }                                                                                           \
    if (nnz_alloc == n_col * n_row && n_col > 0) {                                              \
      for (i64 i = 0; i < nnz_alloc / n_col; ++i) {                                             \
        i64 lo = n_col * i, hi = n_col * (i + 1) - 1;                                           \
        while (lo < hi) {                                                                       \
          T td = out_data[lo]; out_data[lo] = out_data[hi]; out_data[hi] = td;                  \
          i64 ti = out_indices[lo]; out_indices[lo] = out_indices[hi]; out_indices[hi] = ti;    \
          ++lo; --hi;                                                                           \
        }                                                                                       \
      }                                                                                         \
    }                                                                                           \
    free(next_);                                                                                \
    free(sums);                                                                                 \
    return nnz;                                                                                 \
  }

#define DEF_ALL_VAL(I)      \
  DEF_CSR_NDARRAY(f32, I)   \
  DEF_CSR_NDARRAY(f64, I)   \
  DEF_CSR_NDARRAY(i32, I)   \
  DEF_CSR_NDARRAY(i64, I)   \
  DEF_CSC_NDARRAY(f32, I)   \
  DEF_CSC_NDARRAY(f64, I)   \
  DEF_CSC_NDARRAY(i32, I)   \
  DEF_CSC_NDARRAY(i64, I)   \
  DEF_COO_NDARRAY(f32, I)   \
  DEF_COO_NDARRAY(f64, I)   \
  DEF_COO_NDARRAY(i32, I)   \
  DEF_COO_NDARRAY(i64, I)   \
  DEF_CSR_CSR_COUNT(I)      \
  DEF_CSR_CSR(f32, I)       \
  DEF_CSR_CSR(f64, I)       \
  DEF_CSR_CSR(i32, I)       \
  DEF_CSR_CSR(i64, I)

DEF_ALL_VAL(i32)
DEF_ALL_VAL(i64)

/* _match_arrays — reference sparse/numba_backend/_umath.py:70-92: many-to-many sorted join of
 * two non-decreasing int64 key arrays: every (ia, ib) with a[ia] == b[ib], ordered by ia and
 * then ib (the reference re-walks the run of equal b keys for every equal a key through its
 * `match` cursor).  Returns the number of pairs; a_ind/b_ind (may be NULL to only count)
 * receive the pairs. */
i64 oracle_match_arrays(const i64* a, i64 na, const i64* b, i64 nb, i64* a_ind, i64* b_ind) {
  i64 n = 0;
  i64 run = 0; /* first b position not yet proven smaller than the current a key */
  for (i64 ia = 0; ia < na; ++ia) {
    const i64 key = a[ia];
    while (run < nb && b[run] < key) ++run;
    for (i64 ib = run; ib < nb && b[ib] == key; ++ib) {
      if (a_ind) {
        a_ind[n] = ia;
        b_ind[n] = ib;
      }
      ++n;
    }
  }
  return n;
}
