// A1, results of at most 4 columns (N = 1: the matrix-vector product): the "stream" form of `_dot_csr_ndarray`
// (reference sparse/numba_backend/_common.py:720-755) for gfx950.
//
// The product is the CSR triplet's stream - 8 / 12 / 16 bytes per stored element, read once - and nothing else, so the
// kernel is organised around the stream instead of around the rows (spmm_csr.hip's row-vector kernel gives a row to L lanes:
// 4-byte loads at row-aligned addresses, ~1 KB per wave in flight, 4.45 TB/s on config 2's matrix):
//   * every wave owns a contiguous piece of the stream that starts and ends on row boundaries (search in indptr for
//     w * nnz / W, after one round of coalesced loads around the place where an even spread of the elements puts it),
//     walks it in subtiles of 512 elements with 16-byte loads at 16-byte aligned addresses - lane l holds the elements
//     8 l .. 8 l + 7 - and has the next subtile in flight while it works on one (4 KB of index + value per wave at fp32);
//   * B (K x N values) is resident in LDS, one copy per workgroup;
//   * where the rows end inside a subtile comes from a WINDOW of 64 row ends held one per lane (the next window is
//     requested when the current one becomes current, i.e. thousands of elements ahead); the lanes of the window mark
//     the last element of their rows in a 512-bit LDS mask, the element lanes read their eight bits back;
//   * a lane adds its eight products up to the marked elements, a segmented scan over the lanes (DPP; with the wave's carry
//     from the subtile before) completes the rows, and the window lanes PULL their row's sum from the lane that holds
//     its last element (ds_bpermute) and store it: `out` is written in row order, 64 rows per store instruction.
// A row's products are added in a fixed TREE order (in-lane runs of up to eight, then the lanes in ascending order through
// the scan), not k-ascending: deterministic, floating point within rounding of the reference's sum, integers identical -
// the contract of the row-vector kernel this replaces.  SPAMD_EXACT_MULADD / SPAMD_SPMM_ROWGROUP never come here.
#include "common.h"

namespace spamd {

// EPL = consecutive stored elements per lane and subtile: 8 (subtiles of 512 elements) where the kernel then stays inside
// 128 registers - 16 waves per CU -, else 4
template <typename T, typename I, int NV>
constexpr int stream_epl() {
#ifdef SPAMD_STREAM_EPL
  return SPAMD_STREAM_EPL;
#endif
  if (sizeof(T) == 4) return (NV <= 2 || (NV == 3 && sizeof(I) == 4) || (std::is_floating_point<T>::value && sizeof(I) == 4)) ? 8 : 4;
  return (NV == 1 && sizeof(I) == 4) ? 8 : 4;
}
#ifndef SPAMD_STREAM_ABLATE
#define SPAMD_STREAM_ABLATE 0   // timing experiments only (wrong results): 1 = stream loads alone, 2 = + gathers and sums, 3 = + row-end marks, 4 = + scan
#endif
constexpr int ST_ABL = SPAMD_STREAM_ABLATE;
constexpr int ST_MASK_WORDS_MAX = 16;         // the row-end mask of one wave: a bit per element of a subtile
constexpr int ST_LDS_BYTES = 160 * 1024;

// first r in [0, n) with p[r] >= target, n if there is none.  Every lane probes ST_PROBES places per round (the wave 128:
// 10^6 rows take three dependent rounds; a probe instruction touches 64 cache lines, so wider rounds cost the CU's
// vector-memory pipe more than the round they save).  Two targets at once, so that their loads are in flight together.
// This is the FALLBACK: lower_bound_guess2 below answers in one round for matrices whose rows are of similar lengths.
constexpr int ST_PROBES = 2;
constexpr int64_t ST_MAX_ROWS = (int64_t)1 << 28;  // probe offsets are 32-bit byte offsets from the range's start
template <typename I>
__device__ __forceinline__ void lower_bound2(const I* __restrict__ p, int64_t n, int64_t ta, int64_t tb, int lane,
                                             int64_t& ra, int64_t& rb) {
  int64_t lo[2] = {0, 0};
  unsigned len[2] = {(unsigned)n, (unsigned)n};
  const int64_t tg[2] = {ta, tb};
  while (len[0] > 0 || len[1] > 0) {
    unsigned step[2];
    I v[2][ST_PROBES];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      step[q] = (len[q] + 64 * ST_PROBES - 1) / (64 * ST_PROBES);
#pragma unroll
      for (int j = 0; j < ST_PROBES; ++j) {
        // (no load under a branch, here or anywhere in this file: the compiler waits for a predicated load where the
        // branch ends, which puts a memory latency between two loads that could be in flight together)
        const unsigned pos = ((unsigned)lane * ST_PROBES + j) * step[q] + (step[q] - 1);
        const int64_t at = lo[q] + pos < n ? lo[q] + pos : n - 1;
        v[q][j] = p[at];
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (len[q] == 0) continue;
      int fj = ST_PROBES;  // this lane's first probe that qualifies (places past the range qualify by definition)
#pragma unroll
      for (int j = ST_PROBES - 1; j >= 0; --j) {
        const unsigned pos = ((unsigned)lane * ST_PROBES + j) * step[q] + (step[q] - 1);
        if (pos >= len[q] || (int64_t)v[q][j] >= tg[q]) fj = j;
      }
      const unsigned long long bal = __ballot(fj < ST_PROBES);
      if (!bal) {  // every probed element is below the target: none of [lo, lo + len) qualifies
        lo[q] += len[q];
        len[q] = 0;
        continue;
      }
      const int f = __builtin_ctzll(bal);
      const unsigned g = (unsigned)(f * ST_PROBES + __builtin_amdgcn_readlane(fj, f));
      const unsigned first = g * step[q];
      unsigned nl = step[q] - 1;  // the probe of segment g qualifies: the answer is it or one of the step - 1 before it
      if (first + nl > len[q]) nl = len[q] - first;
      lo[q] = uniform(lo[q] + (int64_t)first);
      len[q] = (unsigned)uniform((int)nl);
    }
  }
  ra = lo[0];
  rb = lo[1];
}

// The same two answers in ONE round of coalesced loads when the stored elements are spread evenly over the rows: the
// answer is then near target / nnz * n, and a window of 1024 consecutive entries around that guess (sixteen 256-byte
// loads per target) holds it - at config 2 the cumulative counts wander ~70 rows (sqrt(target) elements) off the straight
// line.  Returns false (both answers unset) when a window misses: the caller searches.
constexpr int ST_GUESS = 16;
template <typename I>
__device__ __forceinline__ bool lower_bound_guess2(const I* __restrict__ p, int64_t n, int64_t nnz, int64_t ta, int64_t tb,
                                                   int lane, int64_t& ra, int64_t& rb) {
  const int64_t tg[2] = {ta, tb};
  int64_t ws[2];
  I v[2][ST_GUESS];
  constexpr int64_t WIN = 64 * ST_GUESS;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double frac = nnz > 0 ? (double)tg[q] / (double)nnz : 0.0;
    int64_t g = (int64_t)(frac * (double)n) - WIN / 2;
    if (g > n - WIN) g = n - WIN;   // (n >= WIN: the caller's condition - every place of a window exists, so the loads
    if (g < 0) g = 0;               //  are one base address + lane + immediate offsets)
    ws[q] = uniform(g);
    const I* wp = p + ws[q] + lane;
#pragma unroll
    for (int j = 0; j < ST_GUESS; ++j) v[q][j] = wp[j * 64];
  }
  bool ok = true;
  int64_t r[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int below = 0;  // loaded entries below the target: they are a prefix of the window (indptr ascends)
#pragma unroll
    for (int j = 0; j < ST_GUESS; ++j) {
      below += __builtin_popcountll(__ballot((int64_t)v[q][j] < tg[q]));
    }
    const int64_t we = ws[q] + WIN;
    // the window holds the answer when it does not begin behind it (its first entry is below the target, or it begins
    // at 0) and does not end before it (an entry at or above the target, or the window reaches the end of the array)
    if ((below == 0 && ws[q] > 0) || (below == (int)(we - ws[q]) && we < n)) ok = false;
    r[q] = ws[q] + below;
  }
  ra = r[0];
  rb = r[1];
  return ok;
}

template <typename T, typename I, int EPL>
struct StreamSub {  // a lane's share of one subtile: EPL consecutive stored elements
  Vec<I, 4> i[EPL / 4];
  Vec<T, 4> v[EPL / 4];
};

// ST_EPL consecutive elements from `base + rel` (elements; base % 4 == 0, arrays 16-byte aligned) with the vector positions
// clamped to `lim` (relative to base): the lanes past the wave's piece re-read its last vector (one cached line) instead of
// branching around the loads - a predicated load with a second arm, or one that keeps the old register contents, makes
// the compiler wait for the data on the spot.  The array's last vector may be partial (nnz % 4 != 0): an aligned 16-byte
// vector that holds one valid element lies inside that element's page, so the load cannot fault; what it holds past the
// end is masked by position.
template <typename T, typename I, int EPL>
__device__ __forceinline__ void stream_load(StreamSub<T, I, EPL>& x, const T* __restrict__ a_data, const I* __restrict__ a_idx,
                                            int64_t base, int rel, int lim) {
#pragma unroll
  for (int h = 0; h < EPL / 4; ++h) {
    int o = rel + 4 * h;
    o = o < lim ? o : lim;
    x.i[h] = *reinterpret_cast<const Vec<I, 4>*>(a_idx + base + o);
    x.v[h] = *reinterpret_cast<const Vec<T, 4>*>(a_data + base + o);
  }
}

template <typename T, int NV>
__device__ __forceinline__ void lds_row(const T* bl, unsigned k, T (&o)[NV]) {
  if constexpr (NV == 3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = bl[k * 3 + c];
  } else {
    const Vec<T, NV> r = *reinterpret_cast<const Vec<T, NV>*>(bl + k * NV);
#pragma unroll
    for (int c = 0; c < NV; ++c) o[c] = r.v[c];
  }
}

template <typename T, int NV>
__device__ __forceinline__ void store_row(T* __restrict__ out, int64_t row, int64_t ldo, const T (&v)[NV], bool vec_ok) {
  constexpr int BYTES = (int)sizeof(T) * NV;
  if constexpr (BYTES == 4 || BYTES == 8 || BYTES == 16) {
    if (vec_ok) {
      hidden_nt_store<T, NV>(out + row * ldo, v);
      return;
    }
  } else if constexpr (BYTES == 32) {
    if (vec_ok) {
      const T a[2] = {v[0], v[1]}, b[2] = {v[2], v[3]};
      hidden_nt_store<T, 2>(out + row * ldo, a);
      hidden_nt_store<T, 2>(out + row * ldo + 2, b);
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const T one[1] = {v[c]};
    hidden_nt_store<T, 1>(out + row * ldo + c, one);
  }
}

// ---- bit-level helpers on 4- and 8-byte values (selects by an all-ones / all-zeros word; DPP moves) ------------------
template <typename T>
struct BitsOf {
  using type = typename std::conditional<sizeof(T) == 4, unsigned, unsigned long long>::type;
};
template <typename T>
__device__ __forceinline__ typename BitsOf<T>::type bits_of(T x) {
  return __builtin_bit_cast(typename BitsOf<T>::type, x);
}
template <typename T>
__device__ __forceinline__ T from_bits(typename BitsOf<T>::type b) {
  return __builtin_bit_cast(T, b);
}
// x where t == 0, zero where t == -1
template <typename T>
__device__ __forceinline__ T keep_unless(T x, int t) {
  using B = typename BitsOf<T>::type;
  return from_bits<T>(bits_of(x) & ~(B)(long long)t);
}
// acc | (x where t == -1)
template <typename T>
__device__ __forceinline__ T or_if(T acc, T x, int t) {
  using B = typename BitsOf<T>::type;
  return from_bits<T>(bits_of(acc) | (bits_of(x) & (B)(long long)t));
}
template <int CTRL, int ROW_MASK, typename T>
__device__ __forceinline__ T dpp_get0(T x) {  // lanes without a source (or outside ROW_MASK) read 0
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, true));
  } else {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
  }
}
template <typename T>
__device__ __forceinline__ T bperm(int byte_addr, T x) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, x)));
  } else {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)(b >> 32));
    return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
  }
}

#ifndef SPAMD_STREAM_WPE
#define SPAMD_STREAM_WPE 4
#endif
#ifdef SPAMD_STREAM_PROF   // timing experiments only: per-wave timestamps (100 MHz wall clock) at four places of the kernel
__device__ unsigned long long st_prof[16384 * 4];
__device__ unsigned long long st_phase[16384 * 8];
#define ST_STAMP(k) do { if (lane == 0 && w < 16384) st_prof[w * 4 + (k)] = wall_clock64(); } while (0)
// cycles (s_memtime) between consecutive probes, summed per wave and phase
#ifdef SPAMD_STREAM_PHASES
#define ST_PROBE(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph_acc[k] += now_ - ph_last; ph_last = now_; } while (0)
#else
#define ST_PROBE(k) do { } while (0)
#endif
#else
#define ST_STAMP(k) do { } while (0)
#define ST_PROBE(k) do { } while (0)
#endif
template <typename T, typename I, int NV, int ST_EPL>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(SPAMD_STREAM_WPE, SPAMD_STREAM_WPE)))
spmm_stream_kernel(int64_t M, int64_t K, const T* __restrict__ a_data, const I* __restrict__ a_idx,
                   const I* __restrict__ a_ptr, const T* __restrict__ b, int64_t ldb, T* __restrict__ out, int64_t ldo,
                   int out_vec_ok, int64_t nnz_hint) {
  constexpr int ST_SUB = 64 * ST_EPL, ST_MASK_WORDS = ST_SUB / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];
  T* bl = reinterpret_cast<T*>(st_smem);
  const int lane = threadIdx.x & (SPAMD_WAVE - 1), wv = threadIdx.x / SPAMD_WAVE, nwv = blockDim.x / SPAMD_WAVE;
  const int64_t bbytes = ((K * NV * (int64_t)sizeof(T)) + 15) & ~(int64_t)15;
  unsigned* mask = reinterpret_cast<unsigned*>(st_smem + bbytes) + wv * ST_MASK_WORDS_MAX;

  // B into LDS, behind the search and the piece's first loads: batches of BB loads in flight per thread, then their
  // writes (a plain copy loop is one load, one wait, one write per trip: twenty L2 latencies in a row at config 2's width)
  constexpr int BB = 8;  // 16-byte vectors (or single values) per thread and batch
  typedef unsigned bvec_t __attribute__((ext_vector_type(4)));
  const int64_t btot = K * NV;
  const bool bcont = ldb == NV && ((uintptr_t)b & 15) == 0;
  // (round 6: NV columns out of a wider B - one pass of a result of more than four columns, spamd_spmm_csr_stream - whose rows
  // are whole 16-byte vectors at 16-byte places: VPR vectors per row, a row every ldb values)
  constexpr int VPR = (NV * (int)sizeof(T)) % 16 == 0 ? NV * (int)sizeof(T) / 16 : 0;
  const bool bstr = !bcont && VPR > 0 && ((uintptr_t)b & 15) == 0 && (ldb * (int64_t)sizeof(T)) % 16 == 0;
  const int64_t nvec = bcont ? btot * (int64_t)sizeof(T) / 16 : bstr ? K * VPR : 0;   // whole vectors of B copied as such
  const int64_t sc0 = nvec * (16 / (int64_t)sizeof(T));               // first value copied one by one
  const int64_t ldb_v = ldb * (int64_t)sizeof(T) / 16;                 // (a row's stride in vectors: strided form)

  // this wave's rows [r_lo, r_hi): the rows whose first element lies in [w, w + 1) * chunk of the stream
  const int64_t W = (int64_t)gridDim.x * nwv, w = (int64_t)blockIdx.x * nwv + wv;
  ST_STAMP(0);
  const int64_t nnz = nnz_hint >= 0 ? nnz_hint : uniform((int64_t)a_ptr[M]);
  const int64_t chunk = (nnz + W - 1) / W;
  int64_t r_lo, r_hi;
#ifdef SPAMD_STREAM_NOGUESS
  if (true)
#else
  if (M + 1 < 64 * ST_GUESS || !lower_bound_guess2<I>(a_ptr, M + 1, nnz, w * chunk, (w + 1) * chunk, lane, r_lo, r_hi))
#endif
    lower_bound2<I>(a_ptr, M + 1, w * chunk, (w + 1) * chunk, lane, r_lo, r_hi);
  if (r_lo > M) r_lo = M;
  if (r_hi > M) r_hi = M;
  if (w == 0) r_lo = 0;
  if (w == W - 1) r_hi = M;
  const bool have = r_lo < r_hi;
  ST_STAMP(1);

  // the piece's ends and the first two windows of row ends: one batch of loads
  const int64_t IMAX = (int64_t)1 << 62;
  auto load_window = [&](int64_t base) -> int64_t {
    const int64_t r = base + lane;
    const int64_t got = (int64_t)a_ptr[r < M ? r + 1 : M];
    return r < r_hi ? got : IMAX;
  };
  int64_t wbase = r_lo;
  I s_raw = 0, e_raw = 0;
  int64_t w0 = IMAX, w1 = IMAX;
  s_raw = a_ptr[r_lo];   // (r_lo, r_hi <= M: valid places whether or not the wave has rows)
  e_raw = a_ptr[r_hi];
  w0 = load_window(wbase);
  w1 = load_window(wbase + 64);

  for (int64_t base = 0; base < nvec; base += (int64_t)BB * blockDim.x) {
    bvec_t hold[BB];
#pragma unroll
    for (int j = 0; j < BB; ++j) {
      const int64_t iv = base + threadIdx.x + (int64_t)j * blockDim.x;
      const int64_t ic = iv < nvec ? iv : nvec - 1;
      int64_t src = ic;
      if constexpr (VPR > 0) src = bstr ? (ic / VPR) * ldb_v + (ic % VPR) : ic;
      hold[j] = reinterpret_cast<const bvec_t*>(b)[src];
    }
#pragma unroll
    for (int j = 0; j < BB; ++j) {
      const int64_t iv = base + threadIdx.x + (int64_t)j * blockDim.x;
      if (iv < nvec) reinterpret_cast<bvec_t*>(bl)[iv] = hold[j];
    }
  }
  for (int64_t base = sc0; base < btot; base += (int64_t)BB * blockDim.x) {  // a strided or unaligned B; a contiguous one's tail
    T one[BB];
#pragma unroll
    for (int j = 0; j < BB; ++j) {
      const int64_t i = base + threadIdx.x + (int64_t)j * blockDim.x;
      const int64_t ic = i < btot ? i : btot - 1;
      one[j] = b[(ic / NV) * ldb + (ic % NV)];
    }
#pragma unroll
    for (int j = 0; j < BB; ++j) {
      const int64_t i = base + threadIdx.x + (int64_t)j * blockDim.x;
      if (i < btot) bl[i] = one[j];
    }
  }
  __syncthreads();
  ST_STAMP(2);
  ST_STAMP(3);
  if (!have) return;

  const int64_t s = uniform((int64_t)s_raw), e = uniform((int64_t)e_raw);
  const bool vec_ok = out_vec_ok != 0;
  if (s == e) {  // nothing stored in these rows
    T z[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) z[c] = T(0);
    for (int64_t r = r_lo + lane; r < r_hi; r += SPAMD_WAVE) store_row<T, NV>(out, r, ldo, z, vec_ok);
    return;
  }

  // windows of row ends: lane j of window `wbase` holds indptr[wbase + 1 + j]; ne = that row holds elements
  auto nonempty_of = [&](int64_t win, int64_t prev) -> bool {
    int64_t st = __shfl_up(win, 1, SPAMD_WAVE);
    if (lane == 0) st = prev;
    return win > st;
  };
  bool ne0 = nonempty_of(w0, s);
  int off = 0;  // rows of the current window already stored

  // per-lane constants
  const int rel = ST_EPL * lane;
  const unsigned* mword = mask + (ST_EPL == 8 ? (lane >> 2) : (lane >> 3));
  const int mshift = ST_EPL == 8 ? (lane & 3) * 8 : (lane & 7) * 4;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;  // the lanes below this one

  const int64_t t_begin = s & ~(int64_t)3;
  const int64_t last_vec = (e - 1) & ~(int64_t)3;  // the last vector that holds an element of the piece
  auto lim_of = [&](int64_t base) -> int {
    const int64_t d = last_vec - base;  // >= -(depth * ST_SUB)
    return d > (int64_t)ST_SUB ? ST_SUB : (int)d;
  };
  // One subtile in flight per wave while the one before it is worked on (4 KB of index + value at fp32; 64 KB per CU): the
  // next subtile is requested into `nxt` at the top of a trip and copied into `cur` at the top of the following one - the
  // copy is where the wave waits for it.  Deeper rings were built and measured (docs/history/r06.md): as compiler loads in
  // an unrolled loop hipcc merges their load sites into one block at the loop's head behind vmcnt(0); as inline-assembly
  // loads with hand-counted waits (three buffers, no copies) the first wave of every SIMD finished its piece in 92 us
  // instead of 115, the fourth as late as before, and the product took 5 % longer.
  StreamSub<T, I, ST_EPL> cur, nxt;
  stream_load<T, I, ST_EPL>(nxt, a_data, a_idx, t_begin, rel, lim_of(t_begin));

  T carry[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) carry[c] = T(0);
  if constexpr (ST_ABL == 5) {  // start-up alone: everything up to the first subtile's loads
    carry[0] = (T)(nxt.v[0].v[0] + (T)nxt.i[0].v[0] + (T)w1 + (T)ne0);
    store_row<T, NV>(out, r_lo + lane, ldo, carry, vec_ok);
    return;
  }

#ifdef SPAMD_STREAM_PHASES
  unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_amdgcn_s_memtime();
#endif
  int64_t t0 = t_begin;
  // one subtile: FULL = every one of its positions belongs to the piece (no masking by position)
  auto subtile = [&](StreamSub<T, I, ST_EPL>& bf, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int64_t t_end = t0 + ST_SUB;
    const unsigned t0p1 = (unsigned)t0 + 1u;
    if constexpr (ST_ABL == 1 || ST_ABL == 2) {
#pragma unroll
      for (int i = 0; i < ST_EPL; ++i) {
        if constexpr (ST_ABL == 1) {
          carry[0] = (T)(carry[0] + bf.v[i / 4].v[i % 4] + (T)bf.i[i / 4].v[i % 4]);
        } else {
          T br[NV];
          lds_row<T, NV>(bl, (unsigned)bf.i[i / 4].v[i % 4], br);
#pragma unroll
          for (int c = 0; c < NV; ++c) carry[c] = mul_add<false>(bf.v[i / 4].v[i % 4], br[c], carry[c]);
        }
      }
      t0 = t_end;
      if (t0 >= e) store_row<T, NV>(out, r_lo + lane, ldo, carry, vec_ok);
      return;
    }
    ST_PROBE(0);   // waiting for the subtile
    // ---- phase A: the window lanes mark the last element of every row that ends in (t0, t_end] ----------------------
    if (lane < ST_MASK_WORDS) mask[lane] = 0;
    const bool in = lane >= off && w0 <= t_end;
    const unsigned q = (unsigned)w0 - t0p1;  // position of the row's last element in the subtile (low words suffice)
    if (in && ne0) atomicOr(&mask[q >> 5], 1u << (q & 31));
    const int cnt = __builtin_popcountll(__ballot(in));
    const bool exhausted = off + cnt == 64 && wbase + 64 < r_hi;
    if (exhausted) {  // rows of later windows end here as well (rows shorter than ~8 elements, runs of empty rows)
      int64_t a0 = w1, a_base = wbase + 64, a_prev = wave_bcast(w0, 63);
      while (true) {
        const bool ain = a0 <= t_end;
        const bool ane = nonempty_of(a0, a_prev);
        const unsigned aq = (unsigned)a0 - t0p1;
        if (ain && ane) atomicOr(&mask[aq >> 5], 1u << (aq & 31));
        if (__builtin_popcountll(__ballot(ain)) == 64 && a_base + 64 < r_hi) {
          a_prev = wave_bcast(a0, 63);
          a_base += 64;
          a0 = load_window(a_base);
          continue;
        }
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned m = (*mword >> mshift) & ((1u << ST_EPL) - 1u);
    const unsigned lowbit = m & (0u - m);

    ST_PROBE(1);   // phase A + mask read
    // ---- the lane's products: running sums that restart behind a marked element ------------------------------------
    int rlo = 0, rhi = ST_SUB;
    if constexpr (!FULL) {
      const int64_t dl = s - t0, dh = e - t0;
      rlo = dl > 0 ? (int)dl : 0;
      rhi = dh < (int64_t)ST_SUB ? (int)dh : ST_SUB;
    }
    T a[ST_EPL][NV], prev[NV], head[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      prev[c] = T(0);
      head[c] = T(0);
    }
#pragma unroll
    for (int i = 0; i < ST_EPL; ++i) {
      const bool ok = FULL || (rel + i >= rlo && rel + i < rhi);
      const unsigned k = ok ? (unsigned)bf.i[i / 4].v[i % 4] : 0u;
      T br[NV];
      lds_row<T, NV>(bl, k, br);
      const int t = (int)(m << (31 - i)) >> 31;        // -1 where element i is the last of a row
      const int t1 = (int)(lowbit << (31 - i)) >> 31;  // -1 where it is the lane's first such element
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const T nx = mul_add<false>(bf.v[i / 4].v[i % 4], br[c], prev[c]);
        a[i][c] = ok ? nx : prev[c];
        prev[c] = keep_unless(a[i][c], t);
        head[c] = or_if(head[c], a[i][c], t1);
      }
    }
    if constexpr (ST_ABL == 3) {
#pragma unroll
      for (int c = 0; c < NV; ++c) carry[c] = (T)(carry[c] + prev[c] + head[c]);
      off += cnt;
      t0 = t_end;
      if (t0 >= e) store_row<T, NV>(out, r_lo + lane, ldo, carry, vec_ok);
      return;
    }
    ST_PROBE(2);   // gathers + in-lane sums
    // ---- segmented scan of the lanes' open tails (DPP): x = the tails back to the nearest marked lane -----------------
    const bool seen = m != 0;
    const unsigned long long G = __ballot(seen);
    const unsigned long long belowS = G & lt_mask;
    // dd: how many lanes back the scan may reach (0 for a marked lane: its tail starts a segment)
    const int mS = belowS ? 63 - __builtin_clzll(belowS) : -1;
    const int dd = seen ? 0 : (belowS ? lane - mS : lane);
    const int lr = lane & 15;
    const bool ok1 = dd >= 1 && lr >= 1, ok2 = dd >= 2 && lr >= 2, ok4 = dd >= 4 && lr >= 4, ok8 = dd >= 8 && lr >= 8;
    const bool oka = dd > lr;            // nothing marked from the row's first lane up to this one: row_bcast15 may come in
    const bool okb = dd > lane - 32;     // the same from lane 32 on (row_bcast31; rows 2 and 3 only, by the row mask)
    T x[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      T v = prev[c], y;
      y = dpp_get0<0x111, 0xf>(v); v = ok1 ? (T)(v + y) : v;   // row_shr:1
      y = dpp_get0<0x112, 0xf>(v); v = ok2 ? (T)(v + y) : v;   // row_shr:2
      y = dpp_get0<0x114, 0xf>(v); v = ok4 ? (T)(v + y) : v;   // row_shr:4
      y = dpp_get0<0x118, 0xf>(v); v = ok8 ? (T)(v + y) : v;   // row_shr:8
      y = dpp_get0<0x142, 0xa>(v); v = oka ? (T)(v + y) : v;   // row_bcast15 -> rows 1, 3
      y = dpp_get0<0x143, 0xc>(v); v = okb ? (T)(v + y) : v;   // row_bcast31 -> rows 2, 3
      x[c] = v;
    }
    const bool first_seg = belowS == 0;  // no marked lane before this one: the carry of the subtile before comes in
    T rone[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      T inc = dpp_get0<0x138, 0xf>(x[c]);  // wave_shr:1 (lane 0 reads 0)
      if (first_seg) inc = (T)(inc + carry[c]);
      rone[c] = (T)(head[c] + inc);
      const T last = wave_bcast(x[c], 63);
      carry[c] = G == 0 ? (T)(carry[c] + last) : last;
    }
    if constexpr (ST_ABL == 4) {
#pragma unroll
      for (int c = 0; c < NV; ++c) carry[c] = (T)(carry[c] + rone[c]);
      off += cnt;
      t0 = t_end;
      if (t0 >= e) store_row<T, NV>(out, r_lo + lane, ldo, carry, vec_ok);
      return;
    }
    ST_PROBE(3);   // scan
    const bool multi = __ballot((m & (m - 1u)) != 0) != 0;  // a lane holds the ends of several rows
    if (multi) {
#pragma unroll
      for (int i = 0; i < ST_EPL; ++i) {
        const bool isff = (lowbit >> i) & 1u;
#pragma unroll
        for (int c = 0; c < NV; ++c) a[i][c] = isff ? rone[c] : a[i][c];
      }
    }

    // ---- phase B: the window lanes pull their rows' sums and store them -----------------------------------------------
    auto pull_store = [&](bool pin, bool pne, unsigned pq, int64_t base) {
      const unsigned qq = (pin && pne) ? pq : 0u;
      const int src = (int)(qq / ST_EPL) * 4;
      T res[NV];
      if (multi) {
        const int el = (int)(qq % ST_EPL);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          res[c] = T(0);
#pragma unroll
          for (int i = 0; i < ST_EPL; ++i) {
            const T got = bperm<T>(src, a[i][c]);
            if (el == i) res[c] = got;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < NV; ++c) res[c] = bperm<T>(src, rone[c]);
      }
      if (pin) {
        if (!pne) {
#pragma unroll
          for (int c = 0; c < NV; ++c) res[c] = T(0);
        }
        store_row<T, NV>(out, base + lane, ldo, res, vec_ok);
      }
    };
    pull_store(in, ne0, q, wbase);
    ST_PROBE(4);   // phase B
    off += cnt;
    if (exhausted) {
      while (true) {
        const int64_t prev_end = wave_bcast(w0, 63);
        wbase += 64;
        w0 = w1;
        w1 = load_window(wbase + 64);
        ne0 = nonempty_of(w0, prev_end);
        const bool bin = w0 <= t_end;
        pull_store(bin, ne0, (unsigned)w0 - t0p1, wbase);
        off = __builtin_popcountll(__ballot(bin));
        if (off == 64 && wbase + 64 < r_hi) continue;
        break;
      }
    }
    t0 = t_end;
  };

#pragma nounroll
  while (true) {
    cur = nxt;
    stream_load<T, I, ST_EPL>(nxt, a_data, a_idx, t0 + ST_SUB, rel, lim_of(t0 + ST_SUB));
    if (t0 >= s && t0 + ST_SUB <= e)
      subtile(cur, std::true_type{});
    else
      subtile(cur, std::false_type{});
    if (t0 >= e) {
      ST_STAMP(3);
#ifdef SPAMD_STREAM_PHASES
      if (lane == 0 && w < 16384)
        for (int k = 0; k < 8; ++k) st_phase[w * 8 + k] = ph_acc[k];
#endif
      return;
    }
  }
}

template <typename T, typename I>
static int launch_stream(int64_t M, int64_t K, int64_t N, const T* a_data, const I* a_idx, const I* a_ptr, const T* b,
                         int64_t ldb, T* out, int64_t ldo, int64_t nnz, unsigned flags, hipStream_t s) {
  const size_t bbytes = ((sizeof(T) * (size_t)N * (size_t)K) + 15) & ~(size_t)15;
  // workgroups of 16 waves, as many per CU as LDS and the kernel's registers (SPAMD_STREAM_WPE waves per SIMD) allow
  int threads = 1024;
  if (((flags >> 16) & 3u) == 2u) threads = 512;  // tuning hint bits 16..17: 2 = workgroups of 8 waves, 3 = of 5
  if (((flags >> 16) & 3u) == 3u) threads = 320;
  const size_t lds = bbytes + (size_t)(threads / 64) * ST_MASK_WORDS_MAX * 4;
  int per_cu = (int)((size_t)ST_LDS_BYTES / lds);
  const int by_regs = SPAMD_STREAM_WPE * 4 * 64 / threads;
  if (per_cu > by_regs) per_cu = by_regs;
  if (per_cu < 1) per_cu = 1;
  int mult = (int)((flags >> 8) & 0xffu);  // tuning hint: workgroups per CU (0 = as many as fit)
  if (mult < 1) mult = per_cu;
  const unsigned blocks = (unsigned)(256 * mult);
  const size_t rowbytes = sizeof(T) * (size_t)N;
  const int vec_ok = (rowbytes == 4 || rowbytes == 8 || rowbytes == 16 || rowbytes == 32) && ldo == N &&
                     (uintptr_t)out % (rowbytes > 16 ? 16 : rowbytes) == 0;
#define SPAMD_ST(NV)                                                                                          \
  case NV: {                                                                                                  \
    auto kern = spmm_stream_kernel<T, I, NV, stream_epl<T, I, NV>()>;                                                                 \
    if (int rc = set_max_dynamic_lds((const void*)kern, ST_LDS_BYTES)) return rc;                             \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, s, M, K, a_data, a_idx, a_ptr, b, ldb, out, ldo, \
                       vec_ok, nnz);                                                                          \
  } break;
  switch (N) {
    SPAMD_ST(1)
    SPAMD_ST(2)
    SPAMD_ST(3)
    case 4:
      if constexpr (sizeof(T) == 4) {
        switch (N) { SPAMD_ST(4) }
        break;
      } else {
        return SPAMD_EINVAL;  // (spamd_spmm_csr_stream_fits says 0)
      }
    default: return SPAMD_EINVAL;
  }
#undef SPAMD_ST
  return launch_status();
}

}  // namespace spamd

#ifdef SPAMD_STREAM_PROF
extern "C" int spamd_stream_prof_read(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(spamd::st_prof), sizeof(unsigned long long) * (size_t)n);
}
extern "C" int spamd_stream_phase_read(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(spamd::st_phase), sizeof(unsigned long long) * (size_t)n);
}
#endif

// Width of one pass: the widest column chunk (at most 4; 3 for 8-byte values - 32-byte rows of B with 64-bit indices do not
// fit the kernel's 128 registers) whose K x width values of B fit the LDS beside the waves' row-end masks; 0 = none.
static int stream_chunk_width(size_t es, int64_t K) {
  const size_t room = (size_t)spamd::ST_LDS_BYTES - 16 * spamd::ST_MASK_WORDS_MAX * 4;
  for (int w = es == 8 ? 3 : 4; w >= 1; --w) {
    const size_t bbytes = ((es * (size_t)w * (size_t)K) + 15) & ~(size_t)15;
    if (bbytes <= room) return w;
  }
  return 0;
}

// 0 = the stream form does not take this product; otherwise the number of PASSES over A it takes: ceil(N / chunk width) - 1
// for results of at most 4 (8-byte values: 3) columns whose B fits the LDS (round 6: a pass costs what A's stream costs
// whatever its width - 0.16-0.18 ms at config 2's matrix, 0.25 with 8-byte values - so two or three passes beat the padded
// panel of the tiled executor, 0.77 / 1.0 ms; the caller decides how many passes are worth it: spamd_stream_passes_worth).
extern "C" int spamd_spmm_csr_stream_fits(int val_dtype, int64_t M, int64_t K, int64_t N, const void* a_data,
                                          const void* a_indices) {
  if (M <= 0 || M >= spamd::ST_MAX_ROWS || K <= 0 || N < 1 || N > 64) return 0;
  const size_t es = (val_dtype == SPAMD_F64 || val_dtype == SPAMD_I64) ? 8 : 4;
  if ((uintptr_t)a_data % 16 || (uintptr_t)a_indices % 16) return 0;
  const int cw = stream_chunk_width(es, K);
  if (cw == 0) return 0;
  return (int)((N + cw - 1) / cw);
}

extern "C" int spamd_spmm_csr_stream(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t N, const void* a_data,
                                     const void* a_indices, const void* a_indptr, const void* b, int64_t ldb, void* out,
                                     int64_t ldo, int64_t nnz, unsigned flags, void* stream) {
  using namespace spamd;
  if (M < 0 || K < 0 || N < 0) return SPAMD_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (!a_indptr || !out || !b || ldo < N || ldb < N) return SPAMD_EINVAL;
  if (!spamd_spmm_csr_stream_fits(val_dtype, M, K, N, a_data, a_indices)) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SPAMD_DISPATCH_VAL(val_dtype, T, {
    SPAMD_DISPATCH_IDX(idx_dtype, I, {
      const int cw = stream_chunk_width(sizeof(T), K);
      for (int64_t c0 = 0; c0 < N; c0 += cw) {       // (one launch per chunk of columns: every pass reads all of A)
        const int64_t nv = N - c0 < cw ? N - c0 : cw;
        if (int rc = launch_stream<T, I>(M, K, nv, (const T*)a_data, (const I*)a_indices, (const I*)a_indptr, (const T*)b + c0, ldb,
                                         (T*)out + c0, ldo, nnz, flags, s))
          return rc;
      }
      return 0;
    })
  })
  return SPAMD_ETYPE;
}
