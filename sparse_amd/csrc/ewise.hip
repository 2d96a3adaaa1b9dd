// A7 / A8: elementwise and grouped-reduce kernels.
//
// A7 (reference `_Elemwise`, sparse/numba_backend/_umath.py:392-751, `_match_arrays` :53-92).
// The reference enumerates presence masks, joins coordinates with argsort + a sorted join, applies
// `func` per mask, concatenates and re-sorts.  On canonical operands (sorted, unique linear keys)
// all of that collapses to ONE sorted-key union: out[k] = func(a[k] or fill_a, b[k] or fill_b),
// then the entries bit-equal to func(fill_a, fill_b) are dropped.  Here:
//   lower_bound_match  : for every key of one operand, its rank in the other + "present there"
//   union_positions    : output slot of every a / b element (a's rank + #unmatched b before it)
//   ewise_binary/unary : the arithmetic on aligned value arrays (HBM-bound streaming)
// A8 (reference `_grouped_reduce` / `ufunc.reduceat`, _coo/core.py:1601-1661):
//   segment_reduce     : one thread per short run (sequential, left to right: bit-identical to
//                        reduceat) or one wave per long run (shuffle tree; fp tolerance).
#include "common.h"

namespace spamd {

#define GRID_STRIDE(i, n)                                                          \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);        \
       i += (int64_t)gridDim.x * blockDim.x)

static inline unsigned grid_for(int64_t n) {
  int64_t b = ceil_div(n, 256);
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// pos[i] = #h < q[i] ; match[i] = (h[pos[i]] == q[i])   (both key arrays sorted, h unique)
__global__ void __launch_bounds__(256) lower_bound_match_kernel(const int64_t* __restrict__ q, int64_t nq,
                                                                const int64_t* __restrict__ h, int64_t nh,
                                                                int64_t* __restrict__ pos,
                                                                int64_t* __restrict__ match) {
  GRID_STRIDE(i, nq) {
    const int64_t key = q[i];
    int64_t lo = 0, hi = nh;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (h[mid] < key) lo = mid + 1; else hi = mid;
    }
    pos[i] = lo;
    match[i] = (lo < nh && h[lo] == key) ? 1 : 0;
  }
}

// Union of two sorted unique key arrays.  ub = exclusive scan of (1 - matchB) (nb+1 entries).
// slot(a_i) = i + ub[posB_i];  slot(b_j) = matchB_j ? slot(a_{posA_j}) : ub_j + posA_j.
__global__ void __launch_bounds__(256) union_positions_kernel(
    const int64_t* __restrict__ ka, int64_t na, const int64_t* __restrict__ posB,
    const int64_t* __restrict__ kb, int64_t nb, const int64_t* __restrict__ posA,
    const int64_t* __restrict__ matchB, const int64_t* __restrict__ ub, int64_t* __restrict__ slotA,
    int64_t* __restrict__ slotB, int64_t* __restrict__ out_keys) {
  GRID_STRIDE(t, na + nb) {
    if (t < na) {
      const int64_t s = t + ub[posB[t]];
      slotA[t] = s;
      out_keys[s] = ka[t];
    } else {
      const int64_t j = t - na;
      const int64_t pa = posA[j];
      if (matchB[j]) {
        // b_j == a_pa ; the unmatched b's before a_pa are exactly the unmatched b's before b_j
        slotB[j] = pa + ub[j];
      } else {
        const int64_t s = ub[j] + pa;
        slotB[j] = s;
        out_keys[s] = kb[j];
      }
    }
  }
}

__global__ void __launch_bounds__(256) invert_flags_kernel(const int64_t* __restrict__ in, int64_t n,
                                                           int64_t* __restrict__ out) {
  GRID_STRIDE(i, n) out[i] = in[i] ? 0 : 1;
}

template <typename U>
__global__ void __launch_bounds__(256) fill_kernel(U* __restrict__ out, int64_t n, U v) {
  GRID_STRIDE(i, n) out[i] = v;
}

// ---- arithmetic -------------------------------------------------------------------------------
enum BinOp {
  B_ADD = 0, B_SUB, B_MUL, B_DIV, B_MAX, B_MIN, B_POW, B_FMAX, B_FMIN,       // T -> T
  B_FLOORDIV, B_REM, B_FMOD, B_COPYSIGN, B_HYPOT, B_ARCTAN2,                  // T -> T (late round 6; the last three: floats)
  B_GT = 32, B_GE, B_LT, B_LE, B_EQ, B_NE, B_LAND, B_LOR, B_LXOR,            // T -> u8
  B_BAND = 64, B_BOR, B_BXOR, B_LSHIFT, B_RSHIFT                              // int -> int
};

// np.floor_divide / np.remainder of floats: NumPy's `npy_divmod` (numpy/_core/src/npymath/npy_math_internal.h.src), statement
// for statement - fmod is exact, so the results are NumPy's bit for bit
template <typename T>
__device__ __forceinline__ void np_divmod(T a, T b, T* floordiv, T* modulus) {
#pragma clang fp contract(off)
  T mod = fmod(a, b);
  if (b == T(0)) {             // (not NaN: b == 0 exactly) - fmod gave NaN, the quotient is a / b
    *modulus = mod;
    *floordiv = a / b;
    return;
  }
  T div = (a - mod) / b;
  if (mod != T(0)) {
    if ((b < T(0)) != (mod < T(0))) {
      mod += b;
      div -= T(1);
    }
  } else {
    mod = copysign(T(0), b);
  }
  T fd;
  if (div != T(0)) {
    fd = floor(div);
    if (div - fd > T(0.5)) fd += T(1);
  } else {
    fd = copysign(T(0), a / b);
  }
  *floordiv = fd;
  *modulus = mod;
}

template <typename T>
__device__ __forceinline__ T np_max(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
  else return a > b ? a : b;
}
template <typename T>
__device__ __forceinline__ T np_min(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return (a != a) ? a : ((b != b) ? b : (a < b ? a : b));
  else return a < b ? a : b;
}

template <typename T>
__device__ __forceinline__ T bin_tt(int op, T a, T b) {
#pragma clang fp contract(off)
  switch (op) {
    case B_ADD: return a + b;
    case B_SUB: return a - b;
    case B_MUL: return a * b;
    case B_DIV:
      if constexpr (std::is_floating_point<T>::value) return a / b;
      else return b == 0 ? T(0) : a / b;
    case B_MAX: return np_max(a, b);
    case B_MIN: return np_min(a, b);
    case B_FMAX:
      if constexpr (std::is_floating_point<T>::value) return (a != a) ? b : ((b != b) ? a : (a > b ? a : b));
      else return a > b ? a : b;
    case B_FMIN:
      if constexpr (std::is_floating_point<T>::value) return (a != a) ? b : ((b != b) ? a : (a < b ? a : b));
      else return a < b ? a : b;
    case B_FLOORDIV:
      if constexpr (std::is_floating_point<T>::value) {
        T q, r;
        np_divmod<T>(a, b, &q, &r);
        return q;
      } else {      // Python's floor division; x // 0 = 0 (NumPy warns and stores 0), MIN // -1 wraps
        if (b == 0) return T(0);
        if (b == T(-1)) return (T)(T(0) - a);
        const T q = a / b;
        return ((a % b != 0) && ((a < 0) != (b < 0))) ? (T)(q - 1) : q;
      }
    case B_REM:
      if constexpr (std::is_floating_point<T>::value) {
        T mod = fmod(a, b);       // `npy_remainder`
        if (b == T(0)) return mod;
        if (mod != T(0)) {
          if ((b < T(0)) != (mod < T(0))) mod += b;
        } else {
          mod = copysign(T(0), b);
        }
        return mod;
      } else {
        if (b == 0 || b == T(-1)) return T(0);
        const T r = a % b;
        return (r != 0 && ((r < 0) != (b < 0))) ? (T)(r + b) : r;
      }
    case B_FMOD:
      if constexpr (std::is_floating_point<T>::value) return fmod(a, b);
      else return (b == 0 || b == T(-1)) ? T(0) : (T)(a % b);
    case B_COPYSIGN:
      if constexpr (std::is_floating_point<T>::value) return copysign(a, b);
      else return a;
    case B_HYPOT:
      if constexpr (std::is_floating_point<T>::value) return hypot(a, b);
      else return a;
    case B_ARCTAN2:
      if constexpr (std::is_floating_point<T>::value) return atan2(a, b);
      else return a;
    case B_POW:
      if constexpr (std::is_same<T, float>::value) return powf(a, b);
      else if constexpr (std::is_same<T, double>::value) return pow(a, b);
      else {  // integer power by squaring: same wrap-around result as b successive multiplications
        T r = 1;
        for (T e = b; e > 0; e >>= 1) {
          if (e & 1) r *= a;
          a *= a;
        }
        return r;
      }
  }
  return T(0);
}

template <typename T>
__device__ __forceinline__ uint8_t bin_tb(int op, T a, T b) {
  switch (op) {
    case B_GT: return a > b;
    case B_GE: return a >= b;
    case B_LT: return a < b;
    case B_LE: return a <= b;
    case B_EQ: return a == b;
    case B_NE: return a != b;
    case B_LAND: return (a != T(0)) && (b != T(0));
    case B_LOR: return (a != T(0)) || (b != T(0));
    case B_LXOR: return (a != T(0)) != (b != T(0));
  }
  return 0;
}

// b_stride / a_stride = 0 broadcasts a scalar held in a 1-element device array
template <typename T>
__global__ void __launch_bounds__(256) binary_tt_kernel(int op, const T* __restrict__ a, int a_stride,
                                                        const T* __restrict__ b, int b_stride, int64_t n,
                                                        T* __restrict__ out) {
  GRID_STRIDE(i, n) {
    const T x = a[i * a_stride], y = b[i * b_stride];
    if (op >= B_BAND) {
      if constexpr (std::is_integral<T>::value) {
        using U = typename std::make_unsigned<T>::type;
        constexpr unsigned long long BITS = sizeof(T) * 8;
        if (op == B_LSHIFT)        // `npy_lshift`: counts outside [0, bits) give 0
          out[i] = (unsigned long long)y < BITS ? (T)((U)x << (unsigned)y) : T(0);
        else if (op == B_RSHIFT)   // `npy_rshift`: ... give the sign
          out[i] = (unsigned long long)y < BITS ? (T)(x >> (unsigned)y) : (x < 0 ? T(-1) : T(0));
        else
          out[i] = op == B_BAND ? (x & y) : (op == B_BOR ? (x | y) : (x ^ y));
      }
    } else {
      out[i] = bin_tt<T>(op, x, y);
    }
  }
}
template <typename T>
__global__ void __launch_bounds__(256) binary_tb_kernel(int op, const T* __restrict__ a, int a_stride,
                                                        const T* __restrict__ b, int b_stride, int64_t n,
                                                        uint8_t* __restrict__ out) {
  GRID_STRIDE(i, n) out[i] = bin_tb<T>(op, a[i * a_stride], b[i * b_stride]);
}

enum UnOp {
  U_NEG = 0, U_ABS, U_SQRT, U_EXP, U_EXPM1, U_LOG, U_LOG1P, U_SIN, U_COS, U_TAN, U_TANH, U_SINH, U_COSH,
  U_ARCSIN, U_ARCTAN, U_FLOOR, U_CEIL, U_RINT, U_TRUNC, U_SIGN, U_SQUARE, U_RECIP, U_POS, U_LOG2, U_LOG10,
  U_EXP2, U_ARCSINH, U_ARCTANH, U_CBRT, U_DEG2RAD, U_RAD2DEG,                     // T -> T
  U_ISNAN = 64, U_ISINF, U_ISFINITE, U_LNOT, U_SIGNBIT                            // T -> u8
};

template <typename T>
__device__ __forceinline__ T un_tt(int op, T x) {
  if constexpr (std::is_floating_point<T>::value) {
    switch (op) {
      case U_NEG: return -x;
      case U_ABS: return fabs(x);
      case U_SQRT: return sqrt(x);
      case U_EXP: return exp(x);
      case U_EXPM1: return expm1(x);
      case U_LOG: return log(x);
      case U_LOG1P: return log1p(x);
      case U_LOG2: return log2(x);
      case U_LOG10: return log10(x);
      case U_EXP2: return exp2(x);
      case U_SIN: return sin(x);
      case U_COS: return cos(x);
      case U_TAN: return tan(x);
      case U_TANH: return tanh(x);
      case U_SINH: return sinh(x);
      case U_COSH: return cosh(x);
      case U_ARCSIN: return asin(x);
      case U_ARCTAN: return atan(x);
      case U_ARCSINH: return asinh(x);
      case U_ARCTANH: return atanh(x);
      case U_CBRT: return cbrt(x);
      case U_FLOOR: return floor(x);
      case U_CEIL: return ceil(x);
      case U_RINT: return rint(x);
      case U_TRUNC: return trunc(x);
      case U_SIGN: return (x != x) ? x : (x > 0 ? T(1) : (x < 0 ? T(-1) : T(0)));
      case U_SQUARE: return x * x;
      case U_RECIP: return T(1) / x;
      case U_POS: return x;
      case U_DEG2RAD: return x * T(0.017453292519943295);
      case U_RAD2DEG: return x * T(57.29577951308232);
    }
  } else {
    switch (op) {
      case U_NEG: return -x;
      case U_ABS: return x < 0 ? -x : x;
      case U_SIGN: return x > 0 ? T(1) : (x < 0 ? T(-1) : T(0));
      case U_SQUARE: return x * x;
      case U_POS: case U_FLOOR: case U_CEIL: case U_RINT: case U_TRUNC: return x;
    }
  }
  return T(0);
}

template <typename T>
__global__ void __launch_bounds__(256) unary_tt_kernel(int op, const T* __restrict__ a, int64_t n, T* __restrict__ out) {
  GRID_STRIDE(i, n) out[i] = un_tt<T>(op, a[i]);
}
template <typename T>
__global__ void __launch_bounds__(256) unary_tb_kernel(int op, const T* __restrict__ a, int64_t n,
                                                       uint8_t* __restrict__ out) {
  GRID_STRIDE(i, n) {
    const T x = a[i];
    uint8_t r = 0;
    if constexpr (std::is_floating_point<T>::value) {
      if (op == U_ISNAN) r = x != x;
      else if (op == U_ISINF) r = (x == x) && ((x - x) != (x - x));
      else if (op == U_ISFINITE) r = (x - x) == (x - x);
      else if (op == U_SIGNBIT) r = signbit(x);
      else if (op == U_LNOT) r = x == T(0);
    } else {
      if (op == U_ISFINITE) r = 1;
      else if (op == U_LNOT) r = x == T(0);
      else if (op == U_SIGNBIT) r = x < 0;
    }
    out[i] = r;
  }
}

// ---- grouped reduce -----------------------------------------------------------------------------
enum RedOp { R_ADD = 0, R_MUL, R_MAX, R_MIN, R_OR, R_AND, R_FMAX, R_FMIN };

template <typename T>
__device__ __forceinline__ T red(int op, T a, T b) {
#pragma clang fp contract(off)
  switch (op) {
    case R_ADD: return a + b;
    case R_MUL: return a * b;
    case R_MAX: return np_max(a, b);
    case R_MIN: return np_min(a, b);
    case R_OR: return (T)((a != T(0)) || (b != T(0)));
    case R_AND: return (T)((a != T(0)) && (b != T(0)));
    case R_FMAX: return bin_tt<T>(B_FMAX, a, b);  // NaN-skipping (np.fmax / np.fmin): nanmax, nanmin
    case R_FMIN: return bin_tt<T>(B_FMIN, a, b);
  }
  return a;
}

// one thread per run: strictly left-to-right, like ufunc.reduceat
template <typename T>
__global__ void __launch_bounds__(256) segreduce_thread_kernel(int op, const T* __restrict__ data, int64_t n,
                                                               const int64_t* __restrict__ heads,
                                                               const int64_t* __restrict__ offs,
                                                               T* __restrict__ out, int64_t* __restrict__ counts) {
  GRID_STRIDE(i, n) {
    if (!heads[i]) continue;
    T acc = data[i];
    int64_t j = i + 1;
    for (; j < n && !heads[j]; ++j) acc = red<T>(op, acc, data[j]);
    out[offs[i]] = acc;
    if (counts) counts[offs[i]] = j - i;
  }
}

// segment starts given explicitly (seg_start[g], g in [0, nseg], seg_start[nseg] = n): one wave per run
template <typename T>
__global__ void __launch_bounds__(256) segreduce_wave_kernel(int op, const T* __restrict__ data,
                                                             const int64_t* __restrict__ seg_start, int64_t nseg,
                                                             T* __restrict__ out, int64_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x / 64);
  for (int64_t g = wave; g < nseg; g += nw) {
    const int64_t s = seg_start[g], e = seg_start[g + 1];
    T acc = T(0);
    bool has = false;
    for (int64_t p = s + lane; p < e; p += 64) {
      acc = has ? red<T>(op, acc, data[p]) : data[p];
      has = true;
    }
    for (int off = 1; off < 64; off <<= 1) {  // in-order pairwise tree keeps lane order
      const T other = __shfl_down(acc, off, 64);
      const bool ohas = __shfl_down((int)has, off, 64) && (lane + off < 64);
      if (ohas) acc = has ? red<T>(op, acc, other) : other;
      has = has || ohas;
    }
    if (lane == 0) {
      out[g] = acc;
      if (counts) counts[g] = e - s;
    }
  }
}

// seg_start[offs[i]] = i for every head i (and seg_start[nseg] = n written by the host wrapper)
__global__ void __launch_bounds__(256) heads_to_starts_kernel(const int64_t* __restrict__ heads,
                                                              const int64_t* __restrict__ offs, int64_t n,
                                                              int64_t nseg, int64_t* __restrict__ seg_start) {
  GRID_STRIDE(i, n + 1) {
    if (i == n) seg_start[nseg] = n;
    else if (heads[i]) seg_start[offs[i]] = i;
  }
}

}  // namespace spamd

using namespace spamd;

#define VAL_SWITCH5(code, T, ...)                                 \
  switch (code) {                                                 \
    case SPAMD_F32: { using T = float; __VA_ARGS__; } break;      \
    case SPAMD_F64: { using T = double; __VA_ARGS__; } break;     \
    case SPAMD_I32: { using T = int32_t; __VA_ARGS__; } break;    \
    case SPAMD_I64: { using T = int64_t; __VA_ARGS__; } break;    \
    case SPAMD_U8: { using T = uint8_t; __VA_ARGS__; } break;     \
    default: return SPAMD_ETYPE;                                  \
  }

extern "C" int spamd_lower_bound_match(int64_t nq, const int64_t* q, int64_t nh, const int64_t* h, int64_t* pos,
                                       int64_t* match, void* stream) {
  if (nq < 0 || nh < 0) return SPAMD_EINVAL;
  if (nq == 0) return 0;
  hipLaunchKernelGGL(lower_bound_match_kernel, dim3(grid_for(nq)), dim3(256), 0, (hipStream_t)stream, q, nq, h, nh,
                     pos, match);
  return launch_status();
}

extern "C" int spamd_invert_flags(int64_t n, const int64_t* in, int64_t* out, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(invert_flags_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, n, out);
  return launch_status();
}

extern "C" int spamd_union_positions(int64_t na, const int64_t* ka, const int64_t* posB, int64_t nb,
                                     const int64_t* kb, const int64_t* posA, const int64_t* matchB,
                                     const int64_t* ub, int64_t* slotA, int64_t* slotB, int64_t* out_keys,
                                     void* stream) {
  if (na < 0 || nb < 0) return SPAMD_EINVAL;
  if (na + nb == 0) return 0;
  hipLaunchKernelGGL(union_positions_kernel, dim3(grid_for(na + nb)), dim3(256), 0, (hipStream_t)stream, ka, na,
                     posB, kb, nb, posA, matchB, ub, slotA, slotB, out_keys);
  return launch_status();
}

extern "C" int spamd_fill(int elem_bytes, int64_t n, void* out, uint64_t value_bits, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  switch (elem_bytes) {
    case 1: hipLaunchKernelGGL(fill_kernel<uint8_t>, dim3(grid_for(n)), dim3(256), 0, s, (uint8_t*)out, n, (uint8_t)value_bits); break;
    case 4: hipLaunchKernelGGL(fill_kernel<uint32_t>, dim3(grid_for(n)), dim3(256), 0, s, (uint32_t*)out, n, (uint32_t)value_bits); break;
    case 8: hipLaunchKernelGGL(fill_kernel<uint64_t>, dim3(grid_for(n)), dim3(256), 0, s, (uint64_t*)out, n, (uint64_t)value_bits); break;
    default: return SPAMD_ETYPE;
  }
  return launch_status();
}

extern "C" int spamd_ewise_binary(int op, int val_dtype, int64_t n, const void* a, int a_is_scalar, const void* b,
                                  int b_is_scalar, void* out, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int as = a_is_scalar ? 0 : 1, bs = b_is_scalar ? 0 : 1;
  const bool to_bool = op >= B_GT && op < B_BAND;
  if (op >= B_BAND && (val_dtype == SPAMD_F32 || val_dtype == SPAMD_F64)) return SPAMD_ETYPE;
  if (op >= B_COPYSIGN && op <= B_ARCTAN2 && val_dtype != SPAMD_F32 && val_dtype != SPAMD_F64) return SPAMD_ETYPE;
  if ((op > B_ARCTAN2 && op < B_GT) || op > B_RSHIFT || op < 0) return SPAMD_EINVAL;
  VAL_SWITCH5(val_dtype, T, {
    if (to_bool)
      hipLaunchKernelGGL(binary_tb_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, op, (const T*)a, as, (const T*)b, bs,
                         n, (uint8_t*)out);
    else
      hipLaunchKernelGGL(binary_tt_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, op, (const T*)a, as, (const T*)b, bs,
                         n, (T*)out);
  })
  return launch_status();
}

// out[i] = mask[i] ? a[i] : b[i]  (np.where on aligned arrays; a / b may be one-element arrays broadcast as scalars);
// values are moved bit-wise
template <typename U>
__global__ void __launch_bounds__(256) select_kernel(const uint8_t* __restrict__ mask, const U* __restrict__ a, int as,
                                                     const U* __restrict__ b, int bs, int64_t n, U* __restrict__ out) {
  GRID_STRIDE(i, n) out[i] = mask[i] ? a[i * as] : b[i * bs];
}

extern "C" int spamd_ewise_select(int elem_bytes, int64_t n, const void* mask_u8, const void* a, int a_is_scalar,
                                  const void* b, int b_is_scalar, void* out, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int as = a_is_scalar ? 0 : 1, bs = b_is_scalar ? 0 : 1;
  switch (elem_bytes) {
    case 1: hipLaunchKernelGGL(select_kernel<uint8_t>, dim3(grid_for(n)), dim3(256), 0, s, (const uint8_t*)mask_u8, (const uint8_t*)a, as, (const uint8_t*)b, bs, n, (uint8_t*)out); break;
    case 4: hipLaunchKernelGGL(select_kernel<uint32_t>, dim3(grid_for(n)), dim3(256), 0, s, (const uint8_t*)mask_u8, (const uint32_t*)a, as, (const uint32_t*)b, bs, n, (uint32_t*)out); break;
    case 8: hipLaunchKernelGGL(select_kernel<uint64_t>, dim3(grid_for(n)), dim3(256), 0, s, (const uint8_t*)mask_u8, (const uint64_t*)a, as, (const uint64_t*)b, bs, n, (uint64_t*)out); break;
    default: return SPAMD_ETYPE;
  }
  return launch_status();
}

extern "C" int spamd_ewise_unary(int op, int val_dtype, int64_t n, const void* a, void* out, void* stream) {
  if (n < 0) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool to_bool = op >= U_ISNAN;
  VAL_SWITCH5(val_dtype, T, {
    if (to_bool)
      hipLaunchKernelGGL(unary_tb_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, op, (const T*)a, n, (uint8_t*)out);
    else
      hipLaunchKernelGGL(unary_tt_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, op, (const T*)a, n, (T*)out);
  })
  return launch_status();
}

// A8 epilogue: fold the implicit fill entries of every group into its reduced value in ONE kernel (reference
// _sparse_array.py:405-422; it was nine elementwise launches and four host-to-device scalar copies):
//   add / multiply   closed form  value (+|*) (n_fill == 0 ? identity : fv*n_fill | fv**n_fill), evaluated in the work
//                    dtype (float64 for floating results, the result dtype for integers) and cast back;
//   other ops        op(value, fv) for the groups that have at least one implicit entry.
template <typename T>
__global__ void __launch_bounds__(256) reduce_fill_kernel(int op, int64_t n, T* __restrict__ vals,
                                                          const int64_t* __restrict__ counts, int64_t n_cols,
                                                          double fv_f, int64_t fv_i) {
#pragma clang fp contract(off)
  GRID_STRIDE(i, n) {
    const int64_t n_fill = n_cols - counts[i];
    if (op == R_ADD || op == R_MUL) {
      if constexpr (std::is_floating_point<T>::value) {
        double contrib;
        if (n_fill == 0) contrib = op == R_ADD ? 0.0 : 1.0;
        else contrib = op == R_ADD ? fv_f * (double)n_fill : pow(fv_f, (double)n_fill);
        const double v = (double)vals[i];
        vals[i] = (T)(op == R_ADD ? v + contrib : v * contrib);
      } else {
        T contrib;
        if (n_fill == 0) contrib = op == R_ADD ? T(0) : T(1);
        else contrib = op == R_ADD ? (T)((T)fv_i * (T)n_fill) : bin_tt<T>(B_POW, (T)fv_i, (T)n_fill);
        vals[i] = op == R_ADD ? (T)(vals[i] + contrib) : (T)(vals[i] * contrib);
      }
    } else if (n_fill != 0) {
      T fv;
      if constexpr (std::is_floating_point<T>::value) fv = (T)fv_f;
      else fv = (T)fv_i;
      vals[i] = red<T>(op, vals[i], fv);
    }
  }
}

// The same with the number of groups still on the device (*n_dev, written by spamd_group_reduce on the same stream), plus
// the count of results that are bit-identical to the result's fill value (the prune of the result container then costs
// nothing when there are none - the usual case - and the host reads both numbers in ONE copy after this launch).
template <typename T>
__global__ void __launch_bounds__(256) reduce_fill_count_kernel(int op, const int64_t* __restrict__ n_dev, T* __restrict__ vals,
                                                                const int64_t* __restrict__ counts, int64_t n_cols, double fv_f,
                                                                int64_t fv_i, uint64_t eq_bits,
                                                                unsigned long long* __restrict__ n_eq) {
#pragma clang fp contract(off)
  const int64_t n = *n_dev;
  unsigned long long same = 0;
  GRID_STRIDE(i, n) {
    const int64_t n_fill = n_cols - counts[i];
    T v = vals[i];
    if (op == R_ADD || op == R_MUL) {
      if constexpr (std::is_floating_point<T>::value) {
        double contrib;
        if (n_fill == 0) contrib = op == R_ADD ? 0.0 : 1.0;
        else contrib = op == R_ADD ? fv_f * (double)n_fill : pow(fv_f, (double)n_fill);
        v = (T)(op == R_ADD ? (double)v + contrib : (double)v * contrib);
      } else {
        T contrib;
        if (n_fill == 0) contrib = op == R_ADD ? T(0) : T(1);
        else contrib = op == R_ADD ? (T)((T)fv_i * (T)n_fill) : bin_tt<T>(B_POW, (T)fv_i, (T)n_fill);
        v = op == R_ADD ? (T)(v + contrib) : (T)(v * contrib);
      }
    } else if (n_fill != 0) {
      T fv;
      if constexpr (std::is_floating_point<T>::value) fv = (T)fv_f;
      else fv = (T)fv_i;
      v = red<T>(op, v, fv);
    }
    vals[i] = v;
    uint64_t b;
    if constexpr (sizeof(T) == 8) b = __builtin_bit_cast(uint64_t, v);
    else if constexpr (sizeof(T) == 4) b = __builtin_bit_cast(uint32_t, v);
    else b = (uint64_t)(uint8_t)v;
    same += b == eq_bits ? 1 : 0;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) same += __shfl_xor(same, d, 64);
  if ((threadIdx.x & 63) == 0 && same) atomicAdd(n_eq, same);
}

extern "C" int spamd_reduce_fill_count(int op, int val_dtype, int64_t n_max, const int64_t* n_dev, void* vals,
                                       const int64_t* counts, int64_t n_cols, double fill_f, int64_t fill_i,
                                       uint64_t result_fill_bits, int64_t* n_eq, void* stream) {
  if (n_max < 0 || op < R_ADD || op > R_FMIN || !n_dev || !n_eq) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n_eq != n_dev + 1) {   // (n_eq = n_dev + 1: the second word of spamd_group_reduce's n_groups, which that call has zeroed)
    hipError_t e = hipMemsetAsync(n_eq, 0, sizeof(int64_t), s);
    if (e != hipSuccess) return (int)e;
  }
  if (n_max == 0) return 0;
  VAL_SWITCH5(val_dtype, T, {
    hipLaunchKernelGGL(reduce_fill_count_kernel<T>, dim3(grid_for(n_max)), dim3(256), 0, s, op, n_dev, (T*)vals, counts,
                       n_cols, fill_f, fill_i, result_fill_bits, reinterpret_cast<unsigned long long*>(n_eq));
  })
  return launch_status();
}

extern "C" int spamd_reduce_fill(int op, int val_dtype, int64_t n, void* vals, const int64_t* counts, int64_t n_cols,
                                 double fill_f, int64_t fill_i, void* stream) {
  if (n < 0 || op < R_ADD || op > R_FMIN) return SPAMD_EINVAL;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  VAL_SWITCH5(val_dtype, T, {
    hipLaunchKernelGGL(reduce_fill_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, op, n, (T*)vals, counts, n_cols, fill_f,
                       fill_i);
  })
  return launch_status();
}

extern "C" int spamd_segment_reduce(int op, int val_dtype, int64_t n, const void* data, const int64_t* heads,
                                    const int64_t* offsets, int64_t nseg, void* out, int64_t* counts,
                                    int64_t* seg_start_ws, void* stream) {
  if (n < 0 || nseg < 0) return SPAMD_EINVAL;
  if (n == 0 || nseg == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool long_runs = seg_start_ws != nullptr && (n / nseg) >= 24;
  VAL_SWITCH5(val_dtype, T, {
    if (long_runs) {
      hipLaunchKernelGGL(heads_to_starts_kernel, dim3(grid_for(n + 1)), dim3(256), 0, s, heads, offsets, n, nseg,
                         seg_start_ws);
      int64_t blocks = ceil_div(nseg, 4);
      if (blocks > 256 * 8) blocks = 256 * 8;
      hipLaunchKernelGGL(segreduce_wave_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, op, (const T*)data,
                         seg_start_ws, nseg, (T*)out, counts);
    } else {
      hipLaunchKernelGGL(segreduce_thread_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, op, (const T*)data, n, heads,
                         offsets, (T*)out, counts);
    }
  })
  return launch_status();
}
