// A4 / A5, row-local form: C = A @ B with both operands compressed by rows (reference `_csr_csr_count_nnz` +
// `_dot_csr_csr`, sparse/numba_backend/_common.py:543-570,639-717).
//
// The global expand-sort-compress of spgemm.hip writes every product to HBM and radix-sorts all of them by
// (row, column): 5 passes over 12 bytes per product.  But the products already arrive grouped by output row, so
// only the columns inside a row need sorting — and a row's products fit in LDS for all but the heaviest rows.
// One workgroup per output row:
//   expand   product p of the row -> (A element e, offset inside B row k_e) by a binary search in an LDS prefix
//            array of the B row lengths; key = column, value = a * b (one rounded multiply);
//   sort     rocprim::block_radix_sort on the column bits (stable LSD: products of equal column stay in the
//            order of A's elements, i.e. the reference's k order);
//   compress every run of equal columns is summed left to right by the thread that owns its head
//            (`sums[j] += ...` in the reference's order: bit-identical), heads are ranked by a block scan.
// Rows are written to a scratch area at their product offset (an upper bound of their length); a second kernel
// packs them once the row lengths are scanned.  Rows are served by size class (workgroup size x items per thread);
// a row whose products (or A elements) exceed the largest class makes the caller fall back to spgemm.hip.
#include <string.h>
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace spamd {

__device__ __forceinline__ int64_t max_seen(const int64_t* p) {  // a stale read only costs a redundant atomic
  return __builtin_nontemporal_load(p);
}

template <typename I>
__global__ void __launch_bounds__(256) spgemm_row_products_kernel(int64_t n_row, const I* __restrict__ a_ptr,
                                                                  const I* __restrict__ a_idx, const I* __restrict__ b_ptr,
                                                                  int64_t* __restrict__ prod, int64_t* __restrict__ maxes) {
  // one wave per row
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t a0 = 0, a1 = 0;
  if (row < n_row) {
    a0 = (int64_t)a_ptr[row];
    a1 = (int64_t)a_ptr[row + 1];
  }
  int64_t s = 0;
  for (int64_t e = a0 + lane; e < a1; e += 64) {
    const int64_t k = (int64_t)a_idx[e];
    s += (int64_t)b_ptr[k + 1] - (int64_t)b_ptr[k];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  // block maxima first: 10^5 same-address atomics from every wave serialise (2 ms); one pair per workgroup does not
  __shared__ int64_t wmax[2][4];
  if (lane == 0) {
    if (row < n_row) prod[row] = s;
    wmax[0][threadIdx.x >> 6] = s;
    wmax[1][threadIdx.x >> 6] = a1 - a0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t m0 = 0, m1 = 0;
    const int64_t first = (int64_t)blockIdx.x * 4;
    for (int w = 0; w < 4 && first + w < n_row; ++w) {
      m0 = wmax[0][w] > m0 ? wmax[0][w] : m0;
      m1 = wmax[1][w] > m1 ? wmax[1][w] : m1;
    }
    if (m0 > max_seen(maxes)) atomicMax(reinterpret_cast<unsigned long long*>(maxes), (unsigned long long)m0);
    if (m1 > max_seen(maxes + 1)) atomicMax(reinterpret_cast<unsigned long long*>(maxes + 1), (unsigned long long)m1);
  }
}

template <int BLOCK, int ITEMS, typename V>
struct RowSortLayout {
#ifndef SPAMD_SPG_RB
#define SPAMD_SPG_RB 0
#define SPAMD_SPG_ALG default_for_radix_sort
#endif
  using sort_t = rocprim::block_radix_sort<int, BLOCK, ITEMS, V, 1, 1, SPAMD_SPG_RB,
                                           rocprim::block_radix_rank_algorithm::SPAMD_SPG_ALG>;
  using scan_t = rocprim::block_scan<int, BLOCK>;
  static constexpr int N = BLOCK * ITEMS;
  // after the sort only the VALUES, one head flag per product and every thread's last column go to LDS (the columns
  // stay in registers): half the footprint of (column, value) pairs, i.e. two workgroups per CU for the big classes
  static constexpr size_t sorted_bytes = (size_t)N * (sizeof(V) + 1) + (size_t)BLOCK * sizeof(int) + 64;
  static constexpr int STAGE = N < 2048 ? N : 2048;  // A elements staged per pass
  static constexpr size_t prefix_bytes = ((size_t)STAGE + 2) * sizeof(int) + (size_t)STAGE * (sizeof(int64_t) + sizeof(V)) + 16;
  static constexpr size_t a(size_t x, size_t y) { return x > y ? x : y; }
  static constexpr size_t bytes = a(a(sizeof(typename sort_t::storage_type), sorted_bytes), prefix_bytes);
};

// rows with lo < products <= hi; V is the value type moved bit-wise except for the multiply / add
template <int BLOCK, int ITEMS, typename V, typename I>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BLOCK == 512 ? 4 : 1, 8)))
spgemm_rowsort_kernel(int64_t n_col, int col_bits, const I* __restrict__ a_ptr, const I* __restrict__ a_idx,
                      const V* __restrict__ a_val, const I* __restrict__ b_ptr, const I* __restrict__ b_idx,
                      const V* __restrict__ b_val, const int64_t* __restrict__ prod_off, int64_t lo, int64_t hi,
                      int* __restrict__ tmp_cols, V* __restrict__ tmp_vals, int64_t* __restrict__ nnz_row) {
#pragma clang fp contract(off)
  using L = RowSortLayout<BLOCK, ITEMS, V>;
  extern __shared__ __attribute__((aligned(16))) char raw[];
  __shared__ typename L::scan_t::storage_type scan_storage;
  const int64_t row = blockIdx.x;
  const int64_t base = prod_off[row];
  const int64_t P = prod_off[row + 1] - base;
  if (P <= lo || P > hi) {
    if (P == 0 && lo < 0 && threadIdx.x == 0) nnz_row[row] = 0;
    return;
  }
  const int tid = threadIdx.x;
  const int64_t a0 = (int64_t)a_ptr[row];
  const int nA = (int)((int64_t)a_ptr[row + 1] - a0);  // <= N (checked by the caller)

  // ---- stage the A row in LDS: prefix[e] = products of the A elements before e (prefix[nA] = P), the start of B
  // row k_e and the A value.  (Every product then needs only LDS lookups and ONE independent pair of global loads;
  // chasing a_idx -> b_ptr -> b_idx per product serialises three memory latencies per item: 81 ms instead of ~10.)
  constexpr int STAGE = L::STAGE;
  int* const prefix = reinterpret_cast<int*>(raw);
  int64_t* const bstart = reinterpret_cast<int64_t*>(raw + (size_t)(STAGE + 1) * sizeof(int) + 4);
  V* const aval = reinterpret_cast<V*>(reinterpret_cast<char*>(bstart) + (size_t)STAGE * sizeof(int64_t));
  int keys[ITEMS];
  V vals[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    keys[j] = (int)n_col;  // sentinel: sorts after every real column
    vals[j] = V(0);
  }
  int chunk_done = 0;
  for (int c0 = 0; c0 < nA; c0 += STAGE) {  // chunks of STAGE A elements (one chunk unless the A row is very long)
    const int cn = nA - c0 < STAGE ? nA - c0 : STAGE;
    // products of the chunks before this one
    // (chunk_base is recomputed from scratch per chunk: rows with more than STAGE elements are rare)
    constexpr int EPT = (STAGE + BLOCK - 1) / BLOCK;
    int len[EPT];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = tid * EPT + j;
      len[j] = 0;
      if (e < cn) {
        const int64_t k = (int64_t)a_idx[a0 + c0 + e];
        const int64_t bs = (int64_t)b_ptr[k];
        len[j] = (int)((int64_t)b_ptr[k + 1] - bs);
        bstart[e] = bs;
        aval[e] = a_val[a0 + c0 + e];
      }
      mine += len[j];
    }
    int before, chunk_total;
    typename L::scan_t().exclusive_scan(mine, before, 0, chunk_total, scan_storage);
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = tid * EPT + j;
      if (e <= cn) prefix[e] = before;
      before += len[j];
    }
    __syncthreads();
    // my products p0 .. p0+ITEMS-1 of the ROW are products (p - done) of this chunk when they fall into it
    const int done = chunk_done;  // products of earlier chunks (block-uniform register)
    const int p0 = tid * ITEMS - done;
    if (p0 + ITEMS > 0 && p0 < chunk_total) {
      int e = 0;
      {
        const int pp = p0 < 0 ? 0 : p0;
        int l = 0, h = cn - 1;
        while (l < h) {
          const int mid = (l + h + 1) >> 1;
          if (prefix[mid] <= pp) l = mid; else h = mid - 1;
        }
        e = l;
      }
      int64_t q[ITEMS];
      V av[ITEMS];
      bool on[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const int p = p0 + j;
        on[j] = p >= 0 && p < chunk_total;
        q[j] = 0;
        av[j] = V(0);
        if (on[j]) {
          while (prefix[e + 1] <= p) ++e;
          q[j] = bstart[e] + (p - prefix[e]);
          av[j] = aval[e];
        }
      }
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        if (on[j]) {
#ifdef SPAMD_SPG_SKIP_LOADS
          keys[j] = (int)(q[j] & 0xffff);
          vals[j] = av[j];
#else
          keys[j] = (int)b_idx[q[j]];
          vals[j] = av[j] * b_val[q[j]];
#endif
        }
      }
    }
    chunk_done += chunk_total;
    __syncthreads();
  }
  __syncthreads();  // prefix[] is dead: the sort reuses the memory

#ifndef SPAMD_SPG_SKIP_SORT
  typename L::sort_t().sort(keys, vals, *reinterpret_cast<typename L::sort_t::storage_type*>(raw), 0, col_bits);
#endif
  __syncthreads();
  V* const sv = reinterpret_cast<V*>(raw);
  int* const lastkey = reinterpret_cast<int*>(raw + (size_t)L::N * sizeof(V));
  unsigned char* const hd = reinterpret_cast<unsigned char*>(lastkey + BLOCK);
  lastkey[tid] = keys[ITEMS - 1];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) sv[tid * ITEMS + j] = vals[j];
  __syncthreads();

  // ---- compress: heads of runs of equal columns
  int nheads = 0;
  bool head[ITEMS];
  {
    int prev = tid ? lastkey[tid - 1] : -1;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int p = tid * ITEMS + j;
      head[j] = p < P && keys[j] != prev;
      prev = keys[j];
      hd[p] = head[j] || p >= P;  // (padding counts as a head: it ends the last run)
      nheads += head[j];
    }
  }
  int rank, total;
  typename L::scan_t().exclusive_scan(nheads, rank, 0, total, scan_storage);  // (its barriers also publish hd[])
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    if (head[j]) {
      const int p = tid * ITEMS + j;
      V acc = vals[j];
      for (int q = p + 1; q < L::N && !hd[q]; ++q) acc = acc + sv[q];
      tmp_cols[base + rank] = keys[j];
      tmp_vals[base + rank] = acc;
      ++rank;
    }
  }
  if (tid == 0) nnz_row[row] = total;
}

// pack the rows: out[indptr[row] + i] = tmp[prod_off[row] + i], i < nnz(row); one workgroup per row
template <typename V>
__global__ void __launch_bounds__(256) spgemm_pack_kernel(const int64_t* __restrict__ prod_off,
                                                          const int64_t* __restrict__ out_ptr, const int* __restrict__ tmp_cols,
                                                          const V* __restrict__ tmp_vals, int64_t* __restrict__ out_idx,
                                                          V* __restrict__ out_val) {
  const int64_t row = blockIdx.x;
  const int64_t src = prod_off[row], dst = out_ptr[row];
  const int64_t n = out_ptr[row + 1] - dst;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    out_idx[dst + i] = (int64_t)tmp_cols[src + i];
    out_val[dst + i] = tmp_vals[src + i];
  }
}

// heavy rows come from the global expand-sort-compress as CSR over ALL rows (src_ptr, int64 columns): put them into
// the scratch at their product offset and record their lengths, so that the pack kernel treats every row alike
template <typename V>
__global__ void __launch_bounds__(256) spgemm_unpack_kernel(const int64_t* __restrict__ rows, const int64_t* __restrict__ src_ptr,
                                                            const int64_t* __restrict__ src_idx, const V* __restrict__ src_val,
                                                            const int64_t* __restrict__ prod_off, int* __restrict__ tmp_cols,
                                                            V* __restrict__ tmp_vals, int64_t* __restrict__ nnz_row) {
  const int64_t row = rows[blockIdx.x];
  const int64_t src = src_ptr[row], dst = prod_off[row];
  const int64_t n = src_ptr[row + 1] - src;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    tmp_cols[dst + i] = (int)src_idx[src + i];
    tmp_vals[dst + i] = src_val[src + i];
  }
  if (threadIdx.x == 0) nnz_row[row] = n;
}

template <int BLOCK, int ITEMS, typename V, typename I>
static int launch_rowsort(int64_t n_row, int64_t n_col, int col_bits, const I* a_ptr, const I* a_idx, const V* a_val,
                          const I* b_ptr, const I* b_idx, const V* b_val, const int64_t* prod_off, int64_t lo, int64_t hi,
                          int* tmp_cols, V* tmp_vals, int64_t* nnz_row, hipStream_t s) {
  using L = RowSortLayout<BLOCK, ITEMS, V>;
  auto kern = &spgemm_rowsort_kernel<BLOCK, ITEMS, V, I>;
  if (L::bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)L::bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)n_row), dim3(BLOCK), L::bytes, s, n_col, col_bits, a_ptr, a_idx, a_val, b_ptr,
                     b_idx, b_val, prod_off, lo, hi, tmp_cols, tmp_vals, nnz_row);
  return launch_status();
}

// size classes: (workgroup, items per thread); the last one bounds what the row-local path accepts
template <typename V>
struct RowClasses {
  static constexpr int64_t c0 = 256 * 2, c1 = 256 * 8, c2 = 512 * 8;
  static constexpr int64_t c3 = sizeof(V) <= 4 ? 1024 * 16 : 1024 * 12;  // 128 KB / 144 KB of sorted (column, value) pairs
};

template <typename V, typename I>
static int rowsort_all(int64_t n_row, int64_t n_col, const I* a_ptr, const I* a_idx, const V* a_val, const I* b_ptr,
                       const I* b_idx, const V* b_val, const int64_t* prod_off, int64_t max_prod, int* tmp_cols,
                       V* tmp_vals, int64_t* nnz_row, hipStream_t s) {
  int col_bits = 1;
  while (((int64_t)1 << col_bits) <= n_col) ++col_bits;  // the sentinel n_col must be representable
  using C = RowClasses<V>;
  int rc = launch_rowsort<256, 2, V, I>(n_row, n_col, col_bits, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val, prod_off, -1, C::c0,
                                        tmp_cols, tmp_vals, nnz_row, s);
  if (rc) return rc;
  if (max_prod > C::c0) {
    rc = launch_rowsort<256, 8, V, I>(n_row, n_col, col_bits, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val, prod_off, C::c0, C::c1,
                                      tmp_cols, tmp_vals, nnz_row, s);
    if (rc) return rc;
  }
  if (max_prod > C::c1) {
    rc = launch_rowsort<512, 8, V, I>(n_row, n_col, col_bits, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val, prod_off, C::c1, C::c2,
                                      tmp_cols, tmp_vals, nnz_row, s);
    if (rc) return rc;
  }
  if (max_prod > C::c2) {
    // 512-thread workgroups with many items per thread: two of them fit a CU (LDS and registers), so the global
    // latencies of one row overlap the sort of another; the 1024-thread class only takes what they cannot hold
    if constexpr (sizeof(V) <= 4) {
      constexpr int64_t c2b = 512 * 24;
      rc = launch_rowsort<512, 24, V, I>(n_row, n_col, col_bits, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val, prod_off, C::c2, c2b,
                                         tmp_cols, tmp_vals, nnz_row, s);
      if (rc) return rc;
      if (max_prod > c2b)
        rc = launch_rowsort<1024, 16, V, I>(n_row, n_col, col_bits, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val, prod_off, c2b,
                                            C::c3, tmp_cols, tmp_vals, nnz_row, s);
    } else {
      constexpr int64_t c2b = 512 * 12;
      rc = launch_rowsort<512, 12, V, I>(n_row, n_col, col_bits, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val, prod_off, C::c2, c2b,
                                         tmp_cols, tmp_vals, nnz_row, s);
      if (rc) return rc;
      if (max_prod > c2b)
        rc = launch_rowsort<1024, 12, V, I>(n_row, n_col, col_bits, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val, prod_off, c2b,
                                            C::c3, tmp_cols, tmp_vals, nnz_row, s);
    }
  }
  return rc;
}

}  // namespace spamd

using namespace spamd;

// prod[row] = number of products of output row `row` (prod has n_row + 1 entries, the last one is zeroed: ready for
// spamd_exclusive_scan); maxes[0] = largest prod, maxes[1] = longest A row (device int64[2], zeroed here).
extern "C" int spamd_spgemm_row_products(int idx_dtype, int64_t n_row, const void* a_indptr, const void* a_indices,
                                         const void* b_indptr, int64_t* prod, int64_t* maxes, void* stream) {
  if (n_row < 0) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(maxes, 0, 2 * sizeof(int64_t), s);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(prod + n_row, 0, sizeof(int64_t), s);
  if (e != hipSuccess) return (int)e;
  if (n_row == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, hipLaunchKernelGGL(spgemm_row_products_kernel<I>, dim3((unsigned)ceil_div(n_row, (int64_t)4)),
                                                      dim3(256), 0, s, n_row, (const I*)a_indptr, (const I*)a_indices,
                                                      (const I*)b_indptr, prod, maxes))
  return launch_status();
}

// Largest number of products (and of A elements) per row the row-local path takes for this value size.
extern "C" int64_t spamd_spgemm_rows_capacity(int val_dtype) {
  return (val_dtype == SPAMD_F32 || val_dtype == SPAMD_I32) ? RowClasses<float>::c3 : RowClasses<double>::c3;
}

// Row-local expand / sort / compress.  prod_off = exclusive scan of prod (n_row + 1); tmp_cols / tmp_vals hold
// prod_off[n_row] entries; nnz_row[n_row + 1] receives the row lengths of C (last entry untouched).
// Rows with more products than spamd_spgemm_rows_capacity are left out (see spamd_spgemm_unpack); rows whose A row is
// longer than the capacity must be left to the global form by the caller as well.  n_col < 2^31 - 1.
extern "C" int spamd_spgemm_rows(int val_dtype, int idx_dtype, int64_t n_row, int64_t n_col, const void* a_indptr,
                                 const void* a_indices, const void* a_data, const void* b_indptr, const void* b_indices,
                                 const void* b_data, const int64_t* prod_off, int64_t max_prod, int* tmp_cols, void* tmp_vals,
                                 int64_t* nnz_row, void* stream) {
  if (n_row < 0 || n_col < 0 || n_col >= 2147483647LL) return SPAMD_EINVAL;
  if (n_row == 0) return 0;
  // rows with more products than the capacity are skipped (nnz_row stays as the caller initialised it): the caller
  // computes them with the global form and hands them over through spamd_spgemm_unpack
  if (max_prod > spamd_spgemm_rows_capacity(val_dtype)) max_prod = spamd_spgemm_rows_capacity(val_dtype);
  hipStream_t s = (hipStream_t)stream;
  SPAMD_DISPATCH_VAL(val_dtype, V, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, return (rowsort_all<V, I>(n_row, n_col, (const I*)a_indptr, (const I*)a_indices,
                                                               (const V*)a_data, (const I*)b_indptr, (const I*)b_indices,
                                                               (const V*)b_data, prod_off, max_prod, tmp_cols, (V*)tmp_vals,
                                                               nnz_row, s)))
  })
  return SPAMD_ETYPE;
}

extern "C" int spamd_spgemm_unpack(int val_dtype, int64_t n_heavy, const int64_t* heavy_rows, const int64_t* src_indptr,
                                   const int64_t* src_indices, const void* src_data, const int64_t* prod_off, int* tmp_cols,
                                   void* tmp_vals, int64_t* nnz_row, void* stream) {
  if (n_heavy < 0) return SPAMD_EINVAL;
  if (n_heavy == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  SPAMD_DISPATCH_VAL(val_dtype, V, hipLaunchKernelGGL(spgemm_unpack_kernel<V>, dim3((unsigned)n_heavy), dim3(256), 0, s, heavy_rows,
                                                      src_indptr, src_indices, (const V*)src_data, prod_off, tmp_cols,
                                                      (V*)tmp_vals, nnz_row))
  return launch_status();
}

extern "C" int spamd_spgemm_pack(int val_dtype, int64_t n_row, const int64_t* prod_off, const int64_t* out_indptr,
                                 const int* tmp_cols, const void* tmp_vals, int64_t* out_indices, void* out_data,
                                 void* stream) {
  if (n_row < 0) return SPAMD_EINVAL;
  if (n_row == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  SPAMD_DISPATCH_VAL(val_dtype, V, hipLaunchKernelGGL(spgemm_pack_kernel<V>, dim3((unsigned)n_row), dim3(256), 0, s, prod_off,
                                                      out_indptr, tmp_cols, (const V*)tmp_vals, out_indices, (V*)out_data))
  return launch_status();
}
