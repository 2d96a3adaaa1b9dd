// N3 / A7: a canonical COO broadcast to a larger shape, materialised in ONE pass, already sorted (round 5).
// Reference: `broadcast_to` (sparse/numba_backend/_umath.py:344-389) and the operand expansion inside `_Elemwise`
// (`_get_expanded_coords_data`, _umath.py:96-167): every stored element is replicated along the broadcast axes.  Rounds 1-4
// built the replicas' keys as an outer sum with eight elementwise / gather passes and then SORTED them (13 C-ABI calls per
// operand: the reference's own broadcast benchmark - (side, 1, side) op (side, side), benchmarks/test_benchmark_coo.py:69-94 -
// took 380-750 us per operation for 10^3 stored elements).
//
// The sorted order of the replicas is known in closed form when the target's axes, left to right, are
//     [B0] [K1] [B1] [K2] [B2]        (B: broadcast axes of the operand, K: axes it really has; any group may be empty)
// - which covers a leading, a middle and a trailing group of broadcast axes.  With b0, k1, b1, k2, b2 the groups' sizes, an
// input key is p1 * k2 + p2 and the replicas in sorted order are: for every beta0, for every RUN of elements with one p1
// (ascending), for every beta1, the run's elements (p2 ascending), each b2 times.  A run [s, e) of the input therefore owns
// the output positions [b1 b2 s, b1 b2 e) of a beta0 block, and a thread finds everything about its output element from its
// own index: the run by a galloping search around element t1 / (b1 b2) (runs are short), then beta1, the element and beta2
// by division.  No scratch, no sort: out_keys ascend by construction.
#include "common.h"

namespace spamd {

template <typename V>
__global__ void __launch_bounds__(256) coo_broadcast_kernel(int64_t n, const int64_t* __restrict__ keys, const V* __restrict__ vals,
                                                           int64_t k1, int64_t b1, int64_t k2, int64_t b2, int64_t n_out,
                                                           int64_t* __restrict__ out_keys, V* __restrict__ out_vals) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_out) return;
  const int64_t per0 = n * b1 * b2;           // outputs of one beta0
  const int64_t beta0 = t / per0, t1 = t - beta0 * per0;
  const int64_t bb = b1 * b2;
  int64_t s = 0, e = n, p1 = 0;
  if (k1 > 1) {
    const int64_t q1 = t1 / bb;               // an element of my run
    p1 = keys[q1] / k2;
    const int64_t lo_key = p1 * k2, hi_key = lo_key + k2;
    // first element of the run: gallop to the left of q1, then bisect
    int64_t hi = q1, step = 1;
    while (hi - step >= 0 && keys[hi - step] >= lo_key) {
      hi -= step;
      step <<= 1;
    }
    int64_t lo = hi - step < 0 ? 0 : hi - step + 1;   // keys[lo - 1] < lo_key (or lo == 0), keys[hi] >= lo_key
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] >= lo_key) hi = mid;
      else lo = mid + 1;
    }
    s = lo;
    // one past its last element: gallop to the right
    lo = q1;
    step = 1;
    while (lo + step < n && keys[lo + step] < hi_key) {
      lo += step;
      step <<= 1;
    }
    hi = lo + step < n ? lo + step : n;                // keys[lo] < hi_key, keys[hi] >= hi_key (or hi == n)
    ++lo;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] >= hi_key) hi = mid;
      else lo = mid + 1;
    }
    e = lo;
  }
  const int64_t cnt = e - s;
  const int64_t t2 = t1 - bb * s;
  const int64_t beta1 = t2 / (cnt * b2), t3 = t2 - beta1 * (cnt * b2);
  const int64_t q = s + t3 / b2, beta2 = t3 - (t3 / b2) * b2;
  const int64_t p2 = keys[q] - p1 * k2;
  out_keys[t] = ((((beta0 * k1 + p1) * b1 + beta1) * k2 + p2) * b2) + beta2;
  out_vals[t] = vals[q];
}

}  // namespace spamd

using namespace spamd;

// keys[n]: sorted, duplicate-free keys of the operand over its own axes (C order), key = p1 * k2 + p2 with p1 < k1, p2 < k2;
// out_keys[n b0 b1 b2] = ((((beta0 k1 + p1) b1 + beta1) k2 + p2) b2 + beta2), ascending; out_vals the replicated values
// (val_bytes 1, 2, 4 or 8, moved bit-wise).  Sizes of empty groups are 1.  The result's size b0 k1 b1 k2 b2 must be < 2^63.
extern "C" int spamd_coo_broadcast(int val_bytes, int64_t n, const int64_t* keys, const void* vals, int64_t b0, int64_t k1, int64_t b1,
                                   int64_t k2, int64_t b2, int64_t* out_keys, void* out_vals, void* stream) {
  if (n < 0 || b0 < 1 || k1 < 1 || b1 < 1 || k2 < 1 || b2 < 1) return SPAMD_EINVAL;
  const int64_t rep = b0 * b1 * b2;
  if (n == 0) return 0;
  if (rep > (((int64_t)1 << 62) / n)) return SPAMD_EINVAL;
  const int64_t n_out = n * rep;
  const int64_t grid = ceil_div(n_out, (int64_t)256);
  if (grid >= ((int64_t)1 << 31)) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
#define BC_GO(V)                                                                                                                 \
  hipLaunchKernelGGL(coo_broadcast_kernel<V>, dim3((unsigned)grid), dim3(256), 0, s, n, keys, (const V*)vals, k1, b1, k2, b2, n_out, \
                     out_keys, (V*)out_vals);                                                                                    \
  return launch_status();
  switch (val_bytes) {
    case 1: { BC_GO(uint8_t) }
    case 2: { BC_GO(uint16_t) }
    case 4: { BC_GO(uint32_t) }
    case 8: { BC_GO(uint64_t) }
    default: return SPAMD_ETYPE;
  }
#undef BC_GO
}
