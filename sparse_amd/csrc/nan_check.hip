// A10: NaN scan over a value array (replaces `nan_check`, reference
// sparse/numba_backend/_common.py:51-69, the full pass `matmul` makes before multiplying).
// HBM-bound streaming read: 16 B per lane per iteration, one atomicOr per wave that saw a NaN.
#include "common.h"

namespace spamd {

// HOST_FLAG: `flag` is pinned (device-mapped) HOST memory that the caller zeroed before the launch; a plain
// system-scope store of 1 needs no PCIe atomics, and the host reads it after waiting on an event recorded behind this
// kernel — so the verdict arrives without draining the product that was queued after the scan.
template <typename T, bool HOST_FLAG = false>
__global__ void __launch_bounds__(256) has_nan_kernel(const T* __restrict__ x, int64_t n, int* flag) {
  constexpr int VEC = 16 / sizeof(T);
  using V = Vec<T, VEC>;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n / VEC;
  bool bad = false;
  const V* xv = reinterpret_cast<const V*>(x);
  for (int64_t i = tid; i < nvec; i += stride) {
    V v = xv[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) bad |= (v.v[e] != v.v[e]);
  }
  for (int64_t i = nvec * VEC + tid; i < n; i += stride) bad |= (x[i] != x[i]);
  if (__any(bad) && (threadIdx.x & (SPAMD_WAVE - 1)) == 0) {
    if constexpr (HOST_FLAG) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else atomicOr(flag, 1);
  }
}

}  // namespace spamd

extern "C" int spamd_has_nan(int val_dtype, int64_t n, const void* data, int* flag, void* stream) {
  using namespace spamd;
  if (n < 0 || !flag) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return 0;
  if (((uintptr_t)data % 16) != 0) return SPAMD_EINVAL;
  int64_t blocks = ceil_div(n, 256 * 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  switch (val_dtype) {
    case SPAMD_F32:
      hipLaunchKernelGGL(has_nan_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s,
                         (const float*)data, n, flag);
      break;
    case SPAMD_F64:
      hipLaunchKernelGGL(has_nan_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s,
                         (const double*)data, n, flag);
      break;
    case SPAMD_I32:
    case SPAMD_I64:
      return 0;  // integers cannot hold NaN
    default:
      return SPAMD_ETYPE;
  }
  return launch_status();
}

// The same scan with the verdict written straight into pinned host memory (`host_flag`: device-accessible pointer of a
// pinned int the CALLER has set to 0).  Asynchronous: the caller records an event behind it and reads the int after
// waiting on that event only.
extern "C" int spamd_has_nan_async(int val_dtype, int64_t n, const void* data, int* host_flag, void* stream) {
  using namespace spamd;
  if (n < 0 || !host_flag) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return 0;
  if (((uintptr_t)data % 16) != 0) return SPAMD_EINVAL;
  int64_t blocks = ceil_div(n, 256 * 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  switch (val_dtype) {
    case SPAMD_F32:
      hipLaunchKernelGGL((has_nan_kernel<float, true>), dim3((unsigned)blocks), dim3(256), 0, s, (const float*)data, n,
                         host_flag);
      break;
    case SPAMD_F64:
      hipLaunchKernelGGL((has_nan_kernel<double, true>), dim3((unsigned)blocks), dim3(256), 0, s, (const double*)data, n,
                         host_flag);
      break;
    case SPAMD_I32:
    case SPAMD_I64:
      return 0;
    default:
      return SPAMD_ETYPE;
  }
  return launch_status();
}

// ---- a few device words to the host without a blocking copy -------------------------------------------------------------------
// `int(t[0])` / `.tolist()` of a device tensor is a stream synchronisation plus a copy command: ~20 us, as much as the
// kernels of a config-1-sized reduction.  Here one thread stores the words into PINNED host memory behind whatever the
// stream holds, then - with release semantics - the call's marker behind them; the host spins on the marker.
static __global__ void deliver_words_kernel(const long long* __restrict__ dev, int n, long long* __restrict__ host,
                                            long long marker) {
  for (int i = 0; i < n; ++i) __hip_atomic_store(&host[i], dev[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&host[n], marker, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int spamd_deliver_words(const int64_t* dev_words, int n, int64_t* host_words, int64_t marker, void* stream) {
  if (n < 1 || n > 16 || !dev_words || !host_words) return SPAMD_EINVAL;
  hipLaunchKernelGGL(deliver_words_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const long long*)dev_words, n,
                     (long long*)host_words, (long long)marker);
  return spamd::launch_status();
}
