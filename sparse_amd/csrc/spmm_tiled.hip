// A1, inspector/executor form (fp32, N a multiple of 128, FMA mode): CSR x dense -> dense with a cached K-tiled
// copy of A (reference loop: sparse/numba_backend/_common.py:744-753).
//
// What round 1 measured: gathering B rows through the vector L1 tops out at
// ~25 TB/s (2.05 ms at config 2).  LDS is 4-5x faster, but B (K x 512 B) does not fit, so B is streamed
// through LDS in K-tiles and the partial sums of a row must live in registers from tile to tile.
// Building each tile's element list on the fly (cursors, windows, readlanes) costs more than the
// gather it replaces, so the list building is hoisted out of the multiply:
//
//   inspector (once per matrix, cached on the container):
//     elements are re-ordered by (35-row group g, K-tile t, row, column) and written as a stream of
//     64-byte BLOCKS of eight (d0, d1) entries, d0 = LDS row offset of the column-in-tile | (2 + 2*row-in-group),
//     d1 = value bits; every (g, t) list is padded to whole blocks with d0 = d1 = 0 (those land in a
//     junk accumulator).  blk_off[g*ntiles + t] = first block of list (g, t).
//   executor (every multiply; this kernel):
//     a workgroup of 16 waves owns 560 rows; wave w owns row group g and keeps its 35 x 128 partial
//     sums in the fixed VGPR block v[58:127] (lane l: columns 2l, 2l+1 of each row);
//     B tile t (160 x 128 fp32 = 80 KB) is copied to LDS by LDS-DMA (global_load_lds_dwordx4),
//     double-buffered, one barrier per tile;
//     the wave pulls ITS blocks with SCALAR loads (s_load_dwordx16, three blocks in a ring), so an
//     entry arrives already wave-uniform: d0 is at once the LDS row offset (v_and_or_b32 with the
//     per-lane base) and — its low 8 bits — the accumulator index for s_set_gpr_idx_idx; d1 is the
//     scalar multiplicand.  Per entry: 1 VALU address op, 1 ds_read_b64, 1 SALU, 2 v_fma_f32.
//     No readlanes, no per-row code, no divergence.
// The inner loops are generated text (tools/gen_tiled_asm.py -> spmm_tiled_asm.inc); the compiler is
// asked to stay in v0..v21 (amdgpu_num_vgpr; tools/check_tiled_regs.py verifies it never touches v22.. outside the asm).
// Per output element the fused multiply-adds happen in the same k-ascending order as in the row-group
// kernel's FMA mode, so both kernels return bit-identical results (column indices sorted within rows).
#include "common.h"
#include "spmm_tiled_asm.inc"
#include <stdlib.h>
#include <mutex>
#include <set>

namespace spamd {

constexpr int TL_RG = TL_ASM_RG;        // rows per wave (row group)
constexpr int TL_WAVES = TL_ASM_WAVES;  // waves per workgroup
#ifndef SPAMD_TL_EXPERIMENTAL
// the generator is parameterised (TL_RG / TL_WAVES in tools/gen_tiled_asm.py).  64 rows x 8 waves (2 waves per SIMD, 256
// registers) passes the parity tests but is 8 % slower with the same two-data-set pipeline (1.04 vs 0.96 ms at config 2):
// two waves per SIMD do not cover the LDS / scalar latencies.  The shipped geometry is the one the tests run.
// 35 rows per wave (70 accumulator registers: the LDS address of an entry is computed into its own result register, which
// freed the four address temporaries) x 16 waves = 560 rows per B tile: 8.6 % less tile DMA than 32 x 16 and, at config 2,
// 1786 workgroups = 6.98 rounds of the 256 CUs instead of 7.63 (8 rounds).
static_assert(TL_RG == 35 && TL_WAVES == 16, "experimental tiled-SpMM geometry: build with -DSPAMD_TL_EXPERIMENTAL");
#endif
constexpr int TL_KB = TL_ASM_KB;  // B rows per tile (tools/gen_tiled_asm.py: 160 = all of the 160 KB LDS in two buffers)
constexpr int TL_NBUF = 2;       // LDS tile buffers: tile t+1 is in flight while tile t is consumed
                                 // (64-row tiles with 3-5 buffers were measured 30-40 % slower: twice the
                                 // barriers and list heads, 17 % padding; 128-row tiles 6 % slower than 160)
// LDS layout: the two buffers are interleaved in 1 KB units (= one LDS-DMA instruction = two B rows), row r of buffer b
// lives at ((r >> 1) * 2 + b) * 1024 + (r & 1) * 512.  The row part of that address is what an entry carries in d0
// (bits 9 and 11..), the buffer bit (10) and the lane offset (3..8) are the per-lane base: one v_and_or_b32 makes the
// address for any tile height, without the power-of-two buffer size a plain (column << 9) | base would need.
__host__ __device__ constexpr int tl_d0(int lc, int lr) { return ((lc >> 1) << 11) | ((lc & 1) << 9) | (2 + 2 * lr); }
constexpr int TL_BLOCK_INTS = 16; // a stream block is 64 bytes

// Per value type: entries per block and where an entry lives inside its block.
//   float : 8 x (d0, value bits)                                           columns per lane 2 (v_pk_fma_f32), panel 128
//   double: d0 of 5 entries, one pad dword, 5 x (value lo, value hi)       columns per lane 1 (v_fma_f64),    panel  64
// In both cases a B row of the panel is 512 bytes in the LDS tile and a lane reads 8 of them with ds_read_b64.
template <typename T>
struct TlFmt;
template <>
struct TlFmt<float> {
  static constexpr int EPB = 8, PANEL = 128;
  __device__ static void put(int* stream, int64_t entry, int d0, float v) {
    int* b = stream + (entry / EPB) * TL_BLOCK_INTS + (entry % EPB) * 2;
    typedef int int2v __attribute__((ext_vector_type(2)));
    int2v e;
    e.x = d0;
    e.y = __builtin_bit_cast(int, v);
    *reinterpret_cast<int2v*>(b) = e;   // (entries are 8-byte aligned: one store instead of two)
  }
};
template <>
struct TlFmt<double> {
  static constexpr int EPB = 5, PANEL = 64;
  __device__ static void put(int* stream, int64_t entry, int d0, double v) {
    int* b = stream + (entry / EPB) * TL_BLOCK_INTS;
    const int slot = (int)(entry % EPB);
    const long long bits = __builtin_bit_cast(long long, v);
    b[slot] = d0;
    if (slot == 0) b[5] = 0;   // (the unused dword of the block: written so that the stream is the same bytes whoever builds it)
    *reinterpret_cast<long long*>(b + 6 + 2 * slot) = bits;   // (8-byte aligned: block + 24 + 8 * slot bytes)
  }
};
typedef int tl_srd_t __attribute__((ext_vector_type(4)));   // buffer descriptor of B (four SGPRs)
constexpr int TL_TILE = TL_KB * 512;
constexpr int TL_LDS = TL_NBUF * TL_TILE;
constexpr int TL_DMA_PER_TILE = TL_TILE / 16 / (TL_WAVES * 64);  // LDS-DMA instructions per wave per tile
static_assert(TL_LDS <= 160 * 1024 && TL_KB % (2 * TL_WAVES) == 0 && TL_DMA_PER_TILE == TL_ASM_DMA_PER_TILE,
              "tile geometry is baked into gen_tiled_asm.py");
constexpr int TL_SLACK_BLOCKS = 68;  // readable blocks past the end of the stream: the fixed-width line touch (<= 64 lines) + ring over-read

#define GRID_STRIDE(i, n)                                                          \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);        \
       i += (int64_t)gridDim.x * blockDim.x)

// key = ((g*ntiles + t)*RG + lr)*KB + lc   from the CSR (row*K + col) key
__global__ void __launch_bounds__(256) tl_keys_kernel(const int64_t* __restrict__ rc_keys, int64_t nnz, int64_t K,
                                                      int64_t ntiles, int64_t* __restrict__ out) {
  GRID_STRIDE(i, nnz) {
    const int64_t k = rc_keys[i];
    const int64_t row = k / K, col = k - row * K;
    const int64_t g = row / TL_RG, lr = row - g * TL_RG, t = col / TL_KB, lc = col - t * TL_KB;
    out[i] = ((g * ntiles + t) * TL_RG + lr) * TL_KB + lc;
  }
}

// sorted tiled keys -> seg_start[nseg + 1] (first sorted position of every list; lists may be empty)
__global__ void __launch_bounds__(256) tl_seg_start_kernel(const int64_t* __restrict__ keys, int64_t nnz,
                                                           int64_t nseg, int64_t* __restrict__ seg_start) {
  GRID_STRIDE(i, nnz + 1) {
    const int64_t hi = i < nnz ? keys[i] / (TL_RG * TL_KB) : nseg;
    const int64_t lo = i > 0 ? keys[i - 1] / (TL_RG * TL_KB) + 1 : 0;
    for (int64_t s = lo; s <= hi; ++s) seg_start[s] = i;  // lists lo..hi start at (or are empty before) i
  }
}

// nblk[s] = blocks of list s (input of the exclusive scan); nblk[nseg] = 0
__global__ void __launch_bounds__(256) tl_blocks_kernel(const int64_t* __restrict__ seg_start, int64_t nseg, int epb,
                                                        int64_t* __restrict__ nblk) {
  GRID_STRIDE(s, nseg + 1) {
    nblk[s] = s < nseg ? (seg_start[s + 1] - seg_start[s] + epb - 1) / epb : 0;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) tl_pack_kernel(const int64_t* __restrict__ keys, const T* __restrict__ vals,
                                                      int64_t nnz, const int64_t* __restrict__ seg_start,
                                                      const int64_t* __restrict__ blk_off, int* __restrict__ stream) {
  GRID_STRIDE(i, nnz) {
    const int64_t k = keys[i];
    const int64_t lc = k % TL_KB, r1 = k / TL_KB, lr = r1 % TL_RG, s = r1 / TL_RG;
    const int64_t dst = blk_off[s] * TlFmt<T>::EPB + (i - seg_start[s]);
    TlFmt<T>::put(stream, dst, tl_d0((int)lc, (int)lr), vals[i]);
  }
}

// ---- direct inspector: CSR with sorted column indices, at most TL_DIRECT_MAX_TILES tiles ------------------
// The tiled order (g, t, row, column) is a STABLE PARTITION BY TILE of each 32-row group of the CSR order,
// so no sort is needed: one workgroup per group counts (row, tile) runs in LDS, and an element's slot is
//   8 * blk_off[g, t] + (elements of tile t in earlier rows of the group) + (position inside its row's run).
// Two passes over A (count, fill) = ~2 GB of traffic at config 2 instead of a 64-bit radix sort of 10^8 pairs.
constexpr int TL_DIRECT_MAX_TILES = 256;  // LDS: 2 * TL_RG * tiles * 4 B (+ tiles * 4) <= 73 KB (35-row groups)

template <typename I>
__device__ __forceinline__ int tl_row_of(const int64_t* rs, int64_t e) {  // largest lr in [0, TL_RG) with rs[lr] <= e
  static_assert(TL_RG <= 128, "seven bisection steps");
  int lo = 0, hi = TL_RG;   // invariant: rs[lo] <= e < rs[hi] (rs[TL_RG] = end of the group)
#pragma unroll
  for (int it = 0; it < (TL_RG <= 64 ? 6 : 7); ++it) {
    const int mid = (lo + hi) >> 1;
    if (hi - lo > 1) {
      if (rs[mid] <= e) lo = mid; else hi = mid;
    }
  }
  return lo;
}

template <typename I>
__global__ void __launch_bounds__(256) tl_count_kernel(int64_t M, int ntiles, int epb, const I* __restrict__ indices,
                                                       const I* __restrict__ indptr, int64_t* __restrict__ nblk,
                                                       int* __restrict__ flags) {
  __shared__ int cnt[TL_DIRECT_MAX_TILES];
  __shared__ int64_t rs[TL_RG + 1];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * TL_RG;
  if (tid <= TL_RG) {
    const int64_t r = r0 + tid;
    rs[tid] = (int64_t)indptr[r < M ? r : M];
  }
  for (int t = tid; t < ntiles; t += 256) cnt[t] = 0;
  __syncthreads();
  const int64_t e0 = rs[0], e1 = rs[TL_RG];
  bool bad = false;
  for (int64_t e = e0 + tid; e < e1; e += 256) {
    const int64_t c = (int64_t)indices[e];
    atomicAdd(&cnt[(int)(c / TL_KB)], 1);
    if (e > e0 && (int64_t)indices[e - 1] > c && rs[tl_row_of<I>(rs, e)] != e) bad = true;  // not a row start
  }
  if (bad) flags[0] = 1;
  __syncthreads();
  for (int t = tid; t < ntiles; t += 256) nblk[(int64_t)blockIdx.x * ntiles + t] = (cnt[t] + epb - 1) / epb;
}

template <typename I, typename T>
__global__ void __launch_bounds__(256) tl_fill_kernel(int64_t M, int ntiles, const T* __restrict__ vals,
                                                      const I* __restrict__ indices, const I* __restrict__ indptr,
                                                      const int64_t* __restrict__ blk_off, int* __restrict__ stream) {
  extern __shared__ int tl_fill_lds[];  // before[32][ntiles], runstart[32][ntiles] (relative to e0), slot0[ntiles]
  __shared__ int64_t rs[TL_RG + 1];
  int* const before = tl_fill_lds;
  int* const runstart = before + TL_RG * ntiles;
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * TL_RG;
  if (tid <= TL_RG) {
    const int64_t r = r0 + tid;
    rs[tid] = (int64_t)indptr[r < M ? r : M];
  }
  for (int i = tid; i < TL_RG * ntiles; i += 256) before[i] = 0;
  __syncthreads();
  const int64_t e0 = rs[0], e1 = rs[TL_RG];
  for (int64_t e = e0 + tid; e < e1; e += 256) {
    const int t = (int)((int64_t)indices[e] / TL_KB);
    const int lr = tl_row_of<I>(rs, e);
    atomicAdd(&before[lr * ntiles + t], 1);
    if (e == rs[lr] || (int)((int64_t)indices[e - 1] / TL_KB) != t) runstart[lr * ntiles + t] = (int)(e - e0);
  }
  __syncthreads();
  for (int t = tid; t < ntiles; t += 256) {  // counts -> number of tile-t elements in earlier rows of the group
    int run = 0;
    for (int lr = 0; lr < TL_RG; ++lr) {
      const int c = before[lr * ntiles + t];
      before[lr * ntiles + t] = run;
      run += c;
    }
  }
  __syncthreads();
  const int64_t* const goff = blk_off + (int64_t)blockIdx.x * ntiles;
  for (int64_t e = e0 + tid; e < e1; e += 256) {
    const int64_t c = (int64_t)indices[e];
    const int t = (int)(c / TL_KB), lc = (int)(c - (int64_t)t * TL_KB);
    const int lr = tl_row_of<I>(rs, e);
    const int64_t dst = goff[t] * TlFmt<T>::EPB + before[lr * ntiles + t] + ((int)(e - e0) - runstart[lr * ntiles + t]);
    TlFmt<T>::put(stream, dst, tl_d0(lc, lr), vals[e]);
  }
}

// ---- fused inspector: count + scan + fill in ONE pass over A ------------------------------------------------------------
// The two-pass builder reads the indices twice, scans the 1.8 M list sizes in a separate launch, clears the 0.87 GB stream
// with a memset and needs the host to read the total before it can allocate.  Here one workgroup per row group counts its
// (row, tile) runs in LDS, scans its tiles' block counts locally, takes its first block from a CLOSED FORM of the row
// pointers (tl_group_first_block: an upper bound of what the groups before it need, so groups are independent: no ticket,
// no look-back), writes its tiles + 1 entries of blk_off, fills its lists and zeroes their padding entries and the gap
// in front of the next group itself.  The stream is allocated for the upper bound ceil(nnz / EPB) + lists (every list
// wastes less than one block).  state = ONE 64-bit word: non-zero = "a row has unsorted column indices" (the caller then
// takes the key-sort recipe; the groups that met such a row wrote zero entries, see group_bad).
__host__ __device__ inline int64_t tl_group_first_block(int64_t e0, int64_t g, int64_t ntiles, int64_t epb) {
  return (e0 + g * ntiles * (epb - 1) + epb - 1) / epb;
}

// `rowmap` (round 5, balanced layouts): the group's TL_RG rows are rowmap[g * TL_RG + lr] (negative = an unused slot) instead of
// the consecutive rows g * TL_RG + lr, and `vstart[g]` = the stored elements of the groups before g (which the closed-form block
// offsets need: the natural layout reads them off the row pointers).
template <typename I, typename T, bool MAPPED>
__global__ void __launch_bounds__(256) tl_inspect_kernel(int64_t M, int ntiles, int64_t groups, const T* __restrict__ vals,
                                                         const I* __restrict__ indices, const I* __restrict__ indptr,
                                                         unsigned long long* __restrict__ state, int* __restrict__ blk_off,
                                                         int* __restrict__ stream, const int* __restrict__ rowmap,
                                                         const int64_t* __restrict__ vstart) {
  extern __shared__ int tl_fill_lds[];  // before[RG][ntiles], runstart[RG][ntiles] (relative to the row's start), loff[ntiles + 1]
  __shared__ __attribute__((aligned(16))) int64_t rsab[2 * (TL_RG + 1)];   // (first element, end) of every row of the group: adjacent words, one LDS read
  __shared__ int wtot[5];
  __shared__ int group_bad;   // a row of THIS group has unsorted column indices: its lists are written as zeros
  constexpr int EPB = TlFmt<T>::EPB;
  int* const before = tl_fill_lds;
  int* const runstart = before + TL_RG * ntiles;
  int* const loff = runstart + TL_RG * ntiles;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t r0 = g * TL_RG;
  if (tid < TL_RG) {
    if constexpr (MAPPED) {
      const int r = rowmap[r0 + tid];
      rsab[2 * (tid)] = r >= 0 ? (int64_t)indptr[r] : 0;
      rsab[2 * (tid) + 1] = r >= 0 ? (int64_t)indptr[r + 1] : 0;
    } else {
      const int64_t r = r0 + tid;
      rsab[2 * (tid)] = (int64_t)indptr[r < M ? r : M];
      rsab[2 * (tid) + 1] = (int64_t)indptr[r + 1 < M ? r + 1 : M];
    }
  }
  for (int i = tid; i < TL_RG * ntiles; i += 256) before[i] = 0;
  if (tid == 0) group_bad = 0;
  __syncthreads();
  const int64_t e0 = MAPPED ? vstart[g] : rsab[2 * (0)], e1 = MAPPED ? vstart[g + 1] : rsab[2 * (TL_RG - 1) + 1];
  bool bad = false;
  // A wave per row (rows wave-strided: the row in the group is known without a bisection over the row starts).  The first
  // TL_PRE * 64 elements of each of the wave's rows (column and value) are requested up front and stay in registers for
  // the count AND the fill phase: one memory latency per workgroup instead of one per row and phase (a workgroup lives
  // ~90 us when every row waits for its own loads: 1.26 -> see DESIGN.md).  Longer rows continue from memory.
  constexpr int RPW = (TL_RG + 3) / 4;   // rows per wave
  constexpr int TL_PRE = 2;
  const int wv = tid >> 6, lane = tid & 63;
  unsigned cpre[RPW][TL_PRE];
  T vpre[RPW][TL_PRE];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int lr = wv + 4 * i;
    const int64_t ra = lr < TL_RG ? rsab[2 * (lr)] : 0, rb = lr < TL_RG ? rsab[2 * (lr) + 1] : 0;
#pragma unroll
    for (int p = 0; p < TL_PRE; ++p) {
      const int64_t e = ra + lane + 64 * p;
      cpre[i][p] = e < rb ? (unsigned)indices[e] : 0xffffffffu;
      vpre[i][p] = e < rb ? vals[e] : T(0);
    }
  }
  auto count_one = [&](int lr, int64_t ra, int64_t e, unsigned c, unsigned cp, bool row_start) {
    const int t = (int)(c / (unsigned)TL_KB);
    atomicAdd(&before[lr * ntiles + t], 1);
    if (!row_start && cp > c) bad = true;
    if (row_start || (int)(cp / (unsigned)TL_KB) != t) runstart[lr * ntiles + t] = (int)(e - (MAPPED ? ra : e0));   // (natural layout: relative to the group's first element, a scalar)
  };
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int lr = wv + 4 * i;
    if (lr >= TL_RG) break;
    const int64_t ra = rsab[2 * (lr)], rb = rsab[2 * (lr) + 1];
#pragma unroll
    for (int p = 0; p < TL_PRE; ++p) {
      const int64_t e = ra + lane + 64 * p;
      const unsigned c = cpre[i][p];
      // the previous element's column: the lane below, or the last lane of the previous chunk
      unsigned cp = (unsigned)__shfl_up((int)c, 1, 64);
      if (p > 0) {
        const unsigned last = (unsigned)__builtin_amdgcn_readlane((int)cpre[i][p > 0 ? p - 1 : 0], 63);
        if (lane == 0) cp = last;
      }
      if (e < rb) count_one(lr, ra, e, c, cp, p == 0 && lane == 0);
    }
    for (int64_t e = ra + lane + 64 * TL_PRE; e < rb; e += 64)
      count_one(lr, ra, e, (unsigned)indices[e], (unsigned)indices[e - 1], false);
  }
  if (bad) {
    atomicOr(&state[0], 1ull);
    group_bad = 1;
  }
  // (Round 6: the fill phase's stores each stand behind a full `s_waitcnt vmcnt(0)` - the preloaded values are first used under
  // `if (e < rb)`, and a wait inside such a block does not hold for the path around it.  Marking every preloaded register as
  // used here, on every path, removes those waits and the kernel is SLOWER: 0.59 -> 0.71 ms (float32 / int32), 1.18 -> 1.30
  // (float64 / int64) at config 2's size, twice in one session.  A wave's 8-byte stores to 18 different lists, paced by
  // their own completion, evidently suit the memory system better than all of them at once.  Left as it was.)
  __syncthreads();
  // per tile: elements of the tile in earlier rows of the group; blocks of the list
  int nb = 0, cnt_t = 0;
  for (int t = tid; t < ntiles; t += 256) {   // (ntiles <= 256: one tile per thread)
    int c[TL_RG];   // all reads first: one LDS latency instead of one per row
#pragma unroll
    for (int lr = 0; lr < TL_RG; ++lr) c[lr] = before[lr * ntiles + t];
    int run = 0;
#pragma unroll
    for (int lr = 0; lr < TL_RG; ++lr) {
      before[lr * ntiles + t] = run;
      run += c[lr];
    }
    cnt_t = run;
    nb = (run + EPB - 1) / EPB;
  }
  // exclusive scan of nb over the tiles (thread = tile)
  const int wid = wv;
  int x = nb;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wtot[wid] = x;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wid; ++w) woff += wtot[w];
  const int gtotal = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  const int my_off = woff + x - nb;
  // First block of this group, in closed form from the row pointers alone (no scan over the groups, no look-back): the
  // groups before this one hold e0 elements in g * ntiles lists, and a list of c elements takes ceil(c / EPB) <=
  // (c + EPB - 1) / EPB blocks, so they take at most (e0 + g * ntiles * (EPB - 1)) / EPB blocks.  Starting every group at
  // that bound leaves a gap (zero-filled below) of less than one block per list in front of the next group; the lists of
  // a group stay contiguous, and blk_off carries ntiles + 1 entries per group (the last one = end of its last list).
  const int64_t goff = tl_group_first_block(e0, g, ntiles, EPB);
  const int64_t gnext = tl_group_first_block(e1, g + 1, ntiles, EPB);
  if (tid < ntiles) loff[tid] = my_off;
  __syncthreads();
  if (tid < ntiles) blk_off[g * (ntiles + 1) + tid] = (int)(goff + my_off);
  if (tid == 0) blk_off[g * (ntiles + 1) + ntiles] = (int)(goff + gtotal);
  for (int64_t i = (goff + gtotal) * TL_BLOCK_INTS + tid; i < gnext * TL_BLOCK_INTS; i += 256) stream[i] = 0;
  if (group_bad) {
    // With unsorted columns a (row, tile) run is not contiguous in the row: `runstart` holds whichever run wrote last, so
    // the slot arithmetic below could leave this group's range (even the allocation) and would leave other slots
    // unwritten.  The per-tile COUNTS do not depend on the order, so the group's lists are well-formed; they are written
    // as zero entries (harmless to the executor: they accumulate into the junk register pair).  The caller sees state[0]
    // and rebuilds the layout by the key-sort recipe; until then nothing downstream reads garbage.
    for (int64_t i = goff * TL_BLOCK_INTS + tid; i < (goff + gtotal) * TL_BLOCK_INTS; i += 256) stream[i] = 0;
    return;
  }
  // fill (a wave per row again; the preloaded elements come from registers)
  auto fill_one = [&](int lr, int64_t ra, int64_t e, unsigned c, T v) {
    const int t = (int)(c / (unsigned)TL_KB);
    const int lc = (int)(c - (unsigned)t * (unsigned)TL_KB);
    const int64_t dst = (goff + loff[t]) * EPB + before[lr * ntiles + t] + ((int)(e - (MAPPED ? ra : e0)) - runstart[lr * ntiles + t]);
    TlFmt<T>::put(stream, dst, tl_d0(lc, lr), v);
  };
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int lr = wv + 4 * i;
    if (lr >= TL_RG) break;
    const int64_t ra = rsab[2 * (lr)], rb = rsab[2 * (lr) + 1];
#pragma unroll
    for (int p = 0; p < TL_PRE; ++p) {
      const int64_t e = ra + lane + 64 * p;
      if (e < rb) fill_one(lr, ra, e, cpre[i][p], vpre[i][p]);
    }
    for (int64_t e = ra + lane + 64 * TL_PRE; e < rb; e += 64) fill_one(lr, ra, e, (unsigned)indices[e], vals[e]);
  }
  // padding entries of my list (zero d0 and value: they accumulate into the junk register pair)
  if (tid < ntiles) {
    const int64_t first = (goff + my_off) * EPB;
    for (int i = cnt_t; i < nb * EPB; ++i) TlFmt<T>::put(stream, first + i, 0, T(0));
  }
}

__device__ __forceinline__ void tl_dma16(unsigned lds_base, const void* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
}

// Tile phases [t0, te) of a wave (generated asm, tools/gen_tiled_asm.py: `phases`): per phase the list loop
// over the blocks [o(t), o(t+1)) reading B rows from LDS tile t, the wave's share of the LDS-DMA of tile
// t+1 (if t+1 < nfull), the scalar request for the first blocks of list t+1, the line touch of list t+2,
// the DMA wait and the barrier.  o(t) = lane (min(t, ntiles) - obase) of `offreg`; o0..o2 = o(t0..t0+2).
// MODE: 0 one fma per term, 3 separate multiply and add (the reference's arithmetic, bit for bit); 4 / 5 the same
// for float64 (5-entry blocks, one column per lane); 6 int32 values and B in the float32 layout (v_mul_lo_u32 + v_add_u32:
// NumPy's wrap-around int32 arithmetic); 1 no fma, 2 no LDS reads / fma (timing ablations).
template <int MODE>
__device__ __forceinline__ void tl_phases(const int* stream, int t0, int te, int o0, int o1, int o2, int offreg, int obase,
                                          int ntiles, int nfull, int toff, int mask, unsigned m0wave,
                                          unsigned row_step, int voff, tl_srd_t srd, unsigned& soff) {
  const int lane = threadIdx.x & 63;
  const unsigned blo = (unsigned)((uintptr_t)stream & 0xffffffffu), bhi = (unsigned)((uintptr_t)stream >> 32);
  const int lane8 = lane * 8;
#define TL_PHASES_OPERANDS                                                                                        \
  [soff] "+s"(soff)                                                                                               \
  : [blo] "s"(blo), [bhi] "s"(bhi), [t0] "s"(t0), [te] "s"(te), [o0] "s"(o0), [o1] "s"(o1), [o2] "s"(o2),         \
    [obase] "s"(obase), [ntiles] "s"(ntiles), [nfull] "s"(nfull), [m0wave] "s"(m0wave), [step] "s"(row_step),      \
    [toff] "v"(toff), [lane8] "v"(lane8), [mask] "v"(mask), [offreg] "v"(offreg), [voff] "v"(voff), [srd] "s"(srd)  \
  : "memory", "m0", "scc", "vcc", TL_CLOB_SGPR, TL_CLOB_TMP, TL_CLOB_ACC
  if (MODE == 6)
    asm volatile(TL_ASM_PHASES_I32 : TL_PHASES_OPERANDS);
  else if (MODE == 4)
    asm volatile(TL_ASM_PHASES_F64 : TL_PHASES_OPERANDS);
  else if (MODE == 5)
    asm volatile(TL_ASM_PHASES_F64_EXACT : TL_PHASES_OPERANDS);
  else if (MODE == 3)
    asm volatile(TL_ASM_PHASES_EXACT : TL_PHASES_OPERANDS);
  else if (MODE == 1)
    asm volatile(TL_ASM_PHASES_NOFMA : TL_PHASES_OPERANDS);
  else if (MODE == 2)
    asm volatile(TL_ASM_PHASES_NOLDS : TL_PHASES_OPERANDS);
  else
    asm volatile(TL_ASM_PHASES : TL_PHASES_OPERANDS);
#undef TL_PHASES_OPERANDS
}

__device__ __forceinline__ void tl_block_map(unsigned L, int npanels, int nblocks, int& bx, int& by) {
  const unsigned pin = npanels > 1 ? 2u : 1u;
  const unsigned per_chunk = (unsigned)((nblocks + 7) / 8) * 8u * pin;
  const unsigned chunk = L / per_chunk, l = L % per_chunk;
  const unsigned seq = l >> 3;
  // (the unsigned divisions are done on the vector unit: back to SGPRs, the asm blocks take B's descriptor as scalars)
  by = uniform((int)(chunk * pin + seq % pin));
  bx = uniform((int)((seq / pin) * 8u + (l & 7u)));
}

// DBG (timing ablation): 2 = no tile DMA.  MODE: see tl_phases.
template <int DBG, int MODE, typename T>
__global__ void __launch_bounds__(TL_WAVES * 64) __attribute__((amdgpu_num_vgpr(TL_ASM_COMP)))
spmm_tiled_kernel(int64_t M, int64_t K, int ntiles, int touch_lines, const int* __restrict__ stream,
                  const int* __restrict__ blk_off, const T* __restrict__ b, int64_t ldb,
                  T* __restrict__ out, int64_t ldo, int last_cols, int npanels, int nblocks, const int* __restrict__ rowmap) {
  constexpr int PANEL = TlFmt<T>::PANEL;            // columns per workgroup: 512 bytes of every B row
  constexpr int CPL = PANEL / 64;                   // columns per lane
  extern __shared__ __attribute__((aligned(16))) char lds[];  // the only LDS object: starts at LDS byte 0
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = uniform(tid >> 6);
  // Workgroup -> (block of row groups, column panel), PAIRS of panels innermost on one XCD: workgroup L runs on XCD L % 8
  // (observed; only speed depends on it).  The panels are taken two at a time (chunk = L / per_chunk); inside a chunk the
  // L / 8-th workgroup of an XCD takes panel (L / 8) % 2 of row-group block ((L / 8) / 2) * 8 + L % 8: the two panels of one
  // block are dispatched back to back to the same XCD, so the block stream that the first pulls from HBM is an L2 hit for
  // the second (a (blocks, panels) grid runs ALL blocks of panel 0 first: float64 at N = 128 then read its stream twice
  // from HBM, 4.4 GB of fabric reads against 2.2 GB algorithmic, 2.1-2.25 ms; pairs 1.95 ms).  Not more than two: every
  // panel in flight on an XCD is another 5 MB of B competing for its 4 MiB L2 (all eight panels of N = 512 innermost: 5.5
  // against 5.0 ms).
  int bx, by;
  tl_block_map(blockIdx.x, npanels, nblocks, bx, by);
  if (bx >= nblocks || by >= npanels) return;              // (whole rounds of the eight XCDs, whole pairs of panels)
  // my row group (lists exist for every wave of the grid).  A 32-bit value ON PURPOSE (the launcher bounds the grid): as an
  // int64 its (always zero) high half stayed live across the phase loop, and with the balanced layout's store below the
  // compiler ran out of the SGPRs the asm blocks leave it and spilled that half into a VGPR lane - a build that faulted
  // intermittently on the GPU (round 5; tools/check_tiled_regs.py now refuses any SGPR spill in these kernels).
  const int g = bx * TL_WAVES + wv;
  b += (int64_t)by * PANEL;                                // column panel of B and of the result
  out += (int64_t)by * PANEL;

  asm volatile(TL_ASM_ZERO ::: "memory", TL_CLOB_ACC);

  // Tile DMA.  A tile is TL_KB B rows x 512 B; a wave issues TL_KB/32 of its LDS-DMA instructions (1 KB each:
  // instruction j of wave w carries rows 32j + 2w and 32j + 2w + 1).  Full tiles are issued from inside
  // the phase asm through a per-thread source pointer that walks down B one round of rows at a time (`dptr`);
  // the last, partial tile goes through `issue_partial`, rows past K clamped to row K-1 (no entry
  // refers to them).
  const int nfull = DBG == 2 ? 0 : (int)(K / TL_KB);
  const unsigned row_step = (unsigned)((2 * TL_WAVES) * ldb * (int64_t)sizeof(T));  // bytes of B covered by one round of DMA instructions
  const unsigned m0wave = (unsigned)wv * 2048u;  // LDS offset of this wave's first row pair (buffer 0)
  // source of the tile DMA: a buffer descriptor of this column panel of B (wave-uniform: SGPRs), the lane's fixed byte
  // offset inside a round of rows, and a walking scalar offset that is an in/out operand of the asm blocks (state parked
  // in "clobbered" registers between asm blocks is only safe while the compiler happens not to need them).  32-bit offsets:
  // the launcher checks that a panel of B spans less than 4 GB.
  const uint64_t bbase = (uint64_t)(uintptr_t)b;
  tl_srd_t srd;
  srd[0] = (int)(bbase & 0xffffffffu);
  srd[1] = (int)((bbase >> 32) & 0xffffu);          // stride 0
  srd[2] = (int)0xffffffffu;                         // num_records: byte range (the entries never refer past K)
  srd[3] = 0x00020000;                               // raw buffer, 32-bit data format
  const int voff = (int)((int64_t)(tid >> 5) * ldb * (int64_t)sizeof(T)) + (tid & 31) * 16;
  unsigned soff = 0;
  auto issue_partial = [&](int t) {
#pragma unroll
    for (int i = 0; i < TL_DMA_PER_TILE; ++i) {
      const int e = (i * (TL_WAVES * 64) + tid) * 16;  // byte inside the tile (512 bytes per row)
      int64_t r = (int64_t)t * TL_KB + (e >> 9);
      if (r >= K) r = K - 1;
      tl_dma16(((unsigned)(i * TL_WAVES + wv) * 2u + (unsigned)(t & 1)) * 1024u,   // row pair i*16 + wv of buffer t & 1
               reinterpret_cast<const char*>(b + r * ldb) + (e & 511));
    }
  };
  const bool has_partial = DBG != 2 && (int64_t)nfull * TL_KB < K;  // tile `nfull` is the partial one

  if (nfull > 0)
    asm volatile(TL_ASM_TILE0 : [soff] "+s"(soff) : [m0wave] "s"(m0wave), [step] "s"(row_step), [voff] "v"(voff), [srd] "s"(srd) : "memory", "m0", "scc", "vcc", "s89");
  else if (has_partial)
    issue_partial(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // List boundaries of this wave: blk_off[g*ntiles + t], t = 0..ntiles.  64 of them at a time live in one
  // VGPR (lane <-> tile) and are read with v_readlane.  A chunk of phases runs in one asm block; a new
  // chunk starts where the offsets register must be reloaded (every 61 tiles) and at the phase that
  // precedes the partial tile.
  // The block stream is read once, by scalar loads: no hardware prefetcher, and every s_waitcnt on the
  // LDS reads (lgkmcnt(0): SMEM returns out of order) also waits for the scalar load issued a round
  // earlier, so its latency must be an L2 hit (~270 cycles), never HBM (~1-2 us: measured 2.25 ms
  // without this).  Two tile phases ahead, the consuming wave touches the 64-byte lines of that list
  // with one vector load (lane i -> line i, result discarded in v23): HBM -> this XCD's L2.
  // (A further scalar-cache prefetch stage — dummy s_load_dword of the next list's lines — was measured
  // 9 % SLOWER: the scalar memory path takes ~20 cycles per 64-byte request and ~5 per dword request per
  // CU whether it hits or not (tools/micro/smem_lat.hip), so extra requests cost more than the latency they save.)
  // (bits 16.. of `touch_lines`: 1 = blk_off holds ntiles + 1 entries per group — each group's own end — instead of
  // one running array in which a group ends where the next one starts: the one-pass inspector's layout)
  const int* const myoff = blk_off + (int64_t)g * (ntiles + (touch_lines >> 16));
  touch_lines &= 0xffff;
  const int toff = (lane < touch_lines ? lane : touch_lines - 1) * 64;  // byte offset of the line this lane touches
  int t = 0;
  while (t < ntiles) {
    const int obase = t;
    const int q = obase + lane;
    int offreg = myoff[q < ntiles ? q : ntiles];
    asm volatile("" : "+v"(offreg));  // (the compiler's wait for this load also drains the DMA: once per chunk)
    int te = obase + 61 < ntiles ? obase + 61 : ntiles;
    if (has_partial && t < nfull - 1 && nfull - 1 < te) te = nfull - 1;  // phase nfull-1 must start a chunk
    if (has_partial && t == nfull - 1) issue_partial(nfull);
    auto o = [&](int tt) { return wave_bcast(offreg, (tt < ntiles ? tt : ntiles) - obase); };
    const int o0 = o(t), o1 = o(t + 1), o2 = o(t + 2);
    if (t == 0) {  // lists 0 and 1 have not been touched by an earlier phase
      const int n = o2 - o0, l = lane < n ? lane : 0;
      asm volatile("global_load_dword " TL_ASM_TOUCH ", %0, off" ::"v"(reinterpret_cast<const char*>(stream + (int64_t)o0 * 16) + l * 64)
                   : "memory", TL_ASM_TOUCH);
    }
    tl_phases<MODE>(stream, t, te, o0, o1, o2, offreg, obase, ntiles, nfull, toff, (int)0xfffffe00, m0wave, row_step, voff, srd, soff);
    t = te;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // write my rows
  const int64_t row0 = (int64_t)g * TL_RG;
  const int64_t left = M - row0;
  const int nvalid = left <= 0 ? 0 : (left < TL_RG ? (int)left : TL_RG);
  T* const obase_p = out + row0 * ldo + lane * CPL;
  const int64_t stride_bytes = ldo * (int64_t)sizeof(T);
  // a result narrower than a whole number of panels: the last panel stores its first `last_cols` columns only (B is
  // zero-padded to whole panels by the caller, C is not: no padded result, no slice pass afterwards); the store block
  // runs under the branch's exec mask
  int bx2, by2;   // (recomputed from the kernel arguments: nothing of the mapping stays live across the asm blocks, whose
  tl_block_map(blockIdx.x, npanels, nblocks, bx2, by2);   //  scalar operands leave the compiler s0..s35)
  const int ncols = (last_cols > 0 && by2 == npanels - 1) ? last_cols : PANEL;
  if (rowmap) {
    // balanced layout: row j of my group is rowmap[g * TL_RG + j] (negative: an unused slot), see tl_map_build_kernel
    const int* const rm = rowmap + (int64_t)g * TL_RG;
    T* const base_p = out + lane * CPL;
    if (lane * CPL + CPL <= ncols)
      asm volatile(TL_ASM_STORE_PERM
                   :
                   : [lo] "v"((unsigned)((uintptr_t)base_p & 0xffffffffu)), [hi] "v"((unsigned)((uintptr_t)base_p >> 32)),
                     [stride] "s"((unsigned)stride_bytes), [rm] "s"(rm)
                   : "memory", "scc", TL_CLOB_SGPR, TL_CLOB_TMP, TL_CLOB_ACC);
    if constexpr (CPL == 2) {
      if (lane * CPL + 1 == ncols)
        asm volatile(TL_ASM_STORE1_PERM
                     :
                     : [lo] "v"((unsigned)((uintptr_t)base_p & 0xffffffffu)), [hi] "v"((unsigned)((uintptr_t)base_p >> 32)),
                       [stride] "s"((unsigned)stride_bytes), [rm] "s"(rm)
                     : "memory", "scc", TL_CLOB_SGPR, TL_CLOB_TMP);
    }
    return;
  }
  if (lane * CPL + CPL <= ncols)
    asm volatile(TL_ASM_STORE
                 :
                 : [lo] "v"((unsigned)((uintptr_t)obase_p & 0xffffffffu)), [hi] "v"((unsigned)((uintptr_t)obase_p >> 32)),
                   [stride] "s"(stride_bytes), [n] "s"(nvalid)
                 : "memory", "scc", "s36", TL_ASM_BASE, TL_ASM_TOUCH, TL_CLOB_ACC);
  if constexpr (CPL == 2) {
    // an odd width (late round 4; a padded result and a slice pass before): the lane whose pair straddles the end of the
    // row stores its first column alone (the accumulators are still where the loop left them)
    if (lane * CPL + 1 == ncols)
      asm volatile(TL_ASM_STORE1
                   :
                   : [lo] "v"((unsigned)((uintptr_t)obase_p & 0xffffffffu)), [hi] "v"((unsigned)((uintptr_t)obase_p >> 32)),
                     [stride] "s"(stride_bytes), [n] "s"(nvalid)
                   : "memory", "scc", "s36", TL_ASM_BASE, TL_ASM_TOUCH);
  }
}

static int64_t tl_grid_groups(int64_t M) { return ceil_div(ceil_div(M, (int64_t)TL_RG), (int64_t)TL_WAVES) * TL_WAVES; }

}  // namespace spamd

using namespace spamd;

static int tl_epb(int val_dtype) { return val_dtype == SPAMD_F32 ? TlFmt<float>::EPB : (val_dtype == SPAMD_F64 ? TlFmt<double>::EPB : 0); }

extern "C" int spamd_spmm_tiled_params(int val_dtype, int* rows_per_group, int* tile_rows, int* groups_per_block,
                                       int* entries_per_block, int* slack_blocks, int* direct_max_tiles,
                                       int* panel_cols) {
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  if (direct_max_tiles) *direct_max_tiles = TL_DIRECT_MAX_TILES;
  if (rows_per_group) *rows_per_group = TL_RG;
  if (tile_rows) *tile_rows = TL_KB;
  if (groups_per_block) *groups_per_block = TL_WAVES;
  if (entries_per_block) *entries_per_block = tl_epb(val_dtype);
  if (slack_blocks) *slack_blocks = TL_SLACK_BLOCKS;
  if (panel_cols) *panel_cols = val_dtype == SPAMD_F32 ? TlFmt<float>::PANEL : TlFmt<double>::PANEL;
  return 0;
}

static unsigned tl_blocks_for(int64_t n) {
  int64_t blocks = ceil_div(n, (int64_t)256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

extern "C" int spamd_spmm_tiled_keys(int64_t nnz, const int64_t* rowcol_keys, int64_t K, int64_t* tiled_keys,
                                     void* stream) {
  if (nnz < 0 || K <= 0) return SPAMD_EINVAL;
  if (nnz == 0) return 0;
  hipLaunchKernelGGL(tl_keys_kernel, dim3(tl_blocks_for(nnz)), dim3(256), 0, (hipStream_t)stream, rowcol_keys, nnz, K,
                     ceil_div(K, (int64_t)TL_KB), tiled_keys);
  return launch_status();
}

extern "C" int spamd_spmm_tiled_lists(int val_dtype, int64_t nnz, const int64_t* tiled_keys_sorted, int64_t M, int64_t K,
                                      int64_t* seg_start, int64_t* nblk, void* stream) {
  if (nnz < 0 || M < 0 || K <= 0) return SPAMD_EINVAL;
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  const int64_t nseg = tl_grid_groups(M) * ceil_div(K, (int64_t)TL_KB);
  hipLaunchKernelGGL(tl_seg_start_kernel, dim3(tl_blocks_for(nnz + 1)), dim3(256), 0, (hipStream_t)stream,
                     tiled_keys_sorted, nnz, nseg, seg_start);
  hipLaunchKernelGGL(tl_blocks_kernel, dim3(tl_blocks_for(nseg + 1)), dim3(256), 0, (hipStream_t)stream, seg_start,
                     nseg, tl_epb(val_dtype), nblk);
  return launch_status();
}

static int tl_clear_blocks(int64_t total_blocks, int* blocks, hipStream_t s) {
  return (int)hipMemsetAsync(blocks, 0, (size_t)(total_blocks + TL_SLACK_BLOCKS) * TL_BLOCK_INTS * sizeof(int), s);
}

extern "C" int spamd_spmm_tiled_pack(int val_dtype, int64_t nnz, const int64_t* tiled_keys_sorted, const void* vals_sorted,
                                     const int64_t* seg_start, const int64_t* blk_off, int64_t total_blocks,
                                     int* blocks, void* stream) {
  if (nnz < 0 || total_blocks < 0) return SPAMD_EINVAL;
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  int rc = tl_clear_blocks(total_blocks, blocks, (hipStream_t)stream);
  if (rc) return rc;
  if (nnz == 0) return 0;
  if (val_dtype == SPAMD_F32)
    hipLaunchKernelGGL(tl_pack_kernel<float>, dim3(tl_blocks_for(nnz)), dim3(256), 0, (hipStream_t)stream, tiled_keys_sorted,
                       (const float*)vals_sorted, nnz, seg_start, blk_off, blocks);
  else
    hipLaunchKernelGGL(tl_pack_kernel<double>, dim3(tl_blocks_for(nnz)), dim3(256), 0, (hipStream_t)stream, tiled_keys_sorted,
                       (const double*)vals_sorted, nnz, seg_start, blk_off, blocks);
  return launch_status();
}

extern "C" int spamd_spmm_tiled_count(int val_dtype, int idx_dtype, int64_t M, int64_t K, const void* a_indices,
                                      const void* a_indptr, int64_t* nblk, int* flags, void* stream) {
  if (M < 0 || K <= 0) return SPAMD_EINVAL;
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  const int64_t ntiles = ceil_div(K, (int64_t)TL_KB);
  if (ntiles > TL_DIRECT_MAX_TILES) return SPAMD_EINVAL;
  const int64_t groups = tl_grid_groups(M);
  hipError_t e = hipMemsetAsync(flags, 0, sizeof(int), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(nblk + groups * ntiles, 0, sizeof(int64_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (groups == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, hipLaunchKernelGGL(tl_count_kernel<I>, dim3((unsigned)groups), dim3(256), 0,
                                                    (hipStream_t)stream, M, (int)ntiles, tl_epb(val_dtype),
                                                    (const I*)a_indices, (const I*)a_indptr, nblk, flags))
  return launch_status();
}

template <typename I, typename T>
static int tl_launch_fill(int64_t M, int64_t ntiles, const T* a_data, const I* a_indices, const I* a_indptr,
                          const int64_t* blk_off, int* blocks, hipStream_t s) {
  const int lds = (int)(2 * TL_RG * ntiles * sizeof(int));
  auto kern = &tl_fill_kernel<I, T>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tl_grid_groups(M)), dim3(256), lds, s, M, (int)ntiles, a_data, a_indices, a_indptr,
                     blk_off, blocks);
  return launch_status();
}

extern "C" int spamd_spmm_tiled_fill(int val_dtype, int idx_dtype, int64_t M, int64_t K, const void* a_data,
                                     const void* a_indices, const void* a_indptr, const int64_t* blk_off,
                                     int64_t total_blocks, int* blocks, void* stream) {
  if (M < 0 || K <= 0 || total_blocks < 0) return SPAMD_EINVAL;
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  const int64_t ntiles = ceil_div(K, (int64_t)TL_KB);
  if (ntiles > TL_DIRECT_MAX_TILES) return SPAMD_EINVAL;
  int rc = tl_clear_blocks(total_blocks, blocks, (hipStream_t)stream);
  if (rc) return rc;
  if (tl_grid_groups(M) == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    if (val_dtype == SPAMD_F32)
      return tl_launch_fill<I, float>(M, ntiles, (const float*)a_data, (const I*)a_indices, (const I*)a_indptr, blk_off,
                                      blocks, (hipStream_t)stream);
    return tl_launch_fill<I, double>(M, ntiles, (const double*)a_data, (const I*)a_indices, (const I*)a_indptr, blk_off,
                                     blocks, (hipStream_t)stream);
  })
  return SPAMD_ETYPE;
}

template <typename I, typename T>
static int tl_launch_inspect(int64_t M, int64_t ntiles, const T* a_data, const I* a_indices, const I* a_indptr,
                             unsigned long long* state, int* blk_off, int* blocks, hipStream_t s, int64_t map_groups = 0,
                             const int* rowmap = nullptr, const int64_t* vstart = nullptr) {
  const int64_t groups = rowmap ? map_groups : tl_grid_groups(M);
  const int lds = (int)((2 * TL_RG * ntiles + ntiles + 1) * sizeof(int));
  auto kern = rowmap ? &tl_inspect_kernel<I, T, true> : &tl_inspect_kernel<I, T, false>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(256), lds, s, M, (int)ntiles, groups, a_data, a_indices, a_indptr,
                     state, blk_off, blocks, rowmap, vstart);
  return launch_status();
}

// ---- balanced layouts for skewed matrices (round 5) -----------------------------------------------------------------------
// The natural layout gives wave w of a workgroup the 35 CONSECUTIVE rows of its group.  A wave works through its list of a
// tile at ~65 cycles per entry whatever its neighbours do (a chain of scalar-load, LDS and FMA latencies: the 16 waves of a
// workgroup hide each other's, not their own), and every tile ends with a barrier: a tile phase lasts as long as the LONGEST
// of the 16 lists.  With Zipf row lengths (a real graph; every bench row until round 5 was uniform `random`) one wave's lists
// are 9x the mean while its 15 neighbours wait - 3.5 ms for config 2's 10^8 elements instead of 0.85.  A balanced layout
// gives every group a ROW MAP instead:
//   * rows are sorted by length, longest first (stable), and classed against a cap C (a power of two, ~2 x the mean group):
//     longer than C/2 -> a group of its own, C/4 -> two to a group, C/8 -> 4, C/16 -> 8, C/32 -> 16, everything else 35 to a
//     group (the unused slots of a group are -1 and cost a zeroed register pair);
//   * groups are numbered in that order, so the 16 groups of a workgroup hold rows of nearly the same length: every phase's
//     16 lists are equally long, whether they hold 160 entries (full rows) or 20.  (Dealing the groups to the workgroups like
//     cards - every workgroup the same mix - was built first and measured: 3.7 ms; the heaviest list still sets each phase.)
//     Heavy workgroups come first, so the tail of the launch is made of light ones;
//   * the inspector reads a group's rows through the map (`vstart` = stored elements before each group, for the closed-form
//     block offsets), the executor stores them through it (TL_ASM_STORE_PERM).  Nothing else changes: the entries, the lists
//     and the order in which an output element's terms are added are the same, so the product is bit-identical.
constexpr int TL_MAP_CLASSES = 6;
__host__ __device__ inline int tl_map_slots(int c) { return c == 5 ? TL_RG : (1 << c); }
__host__ __device__ inline int tl_map_class(int64_t len, int64_t cap) {
  return len > cap / 2 ? 0 : (len > cap / 4 ? 1 : (len > cap / 8 ? 2 : (len > cap / 16 ? 3 : (len > cap / 32 ? 4 : 5))));
}

// keys[r] = maxlen - length of row r (ascending keys = longest rows first), rows[r] = r; stats[0..5] += rows per class
template <typename I>
__global__ void __launch_bounds__(256) tl_map_stats_kernel(int64_t M, int64_t cap, int64_t maxlen, const I* __restrict__ indptr,
                                                           int64_t* __restrict__ keys, int* __restrict__ rows,
                                                           unsigned long long* __restrict__ stats) {
  __shared__ unsigned cnt[TL_MAP_CLASSES];
  if (threadIdx.x < TL_MAP_CLASSES) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r0 = (int64_t)blockIdx.x * blockDim.x; r0 < M; r0 += stride) {     // (whole waves: the class counts are ballots)
    const int64_t r = r0 + threadIdx.x;
    int c = -1;
    if (r < M) {
      const int64_t len = (int64_t)indptr[r + 1] - (int64_t)indptr[r];
      c = tl_map_class(len, cap);
      keys[r] = len < maxlen ? maxlen - len : 0;
      rows[r] = (int)r;
    }
#pragma unroll
    for (int k = 0; k < TL_MAP_CLASSES; ++k) {
      const unsigned long long m = __ballot(c == k);
      if (lane == 0 && m) atomicAdd(&cnt[k], (unsigned)__popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < TL_MAP_CLASSES && cnt[threadIdx.x]) atomicAdd(&stats[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

// stats[6] = stored elements of the heaviest NATURAL row group (35 consecutive rows): the "is this operand skewed" test, one
// thread per group - all an operand that is not skewed pays for the balanced layouts
template <typename I>
__global__ void __launch_bounds__(256) tl_map_skew_kernel(int64_t M, const I* __restrict__ indptr,
                                                          unsigned long long* __restrict__ stats) {
  const int64_t groups = (M + TL_RG - 1) / TL_RG;
  unsigned long long mine = 0;
  GRID_STRIDE(g, groups) {
    const int64_t r = g * TL_RG, e = r + TL_RG < M ? r + TL_RG : M;
    const unsigned long long load = (unsigned long long)((int64_t)indptr[e] - (int64_t)indptr[r]);
    mine = load > mine ? load : mine;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned long long y = __shfl_xor(mine, d, 64);
    mine = y > mine ? y : mine;
  }
  if ((threadIdx.x & 63) == 0 && mine) atomicMax(&stats[6], mine);
}

struct TlMapPlan {
  int64_t cstart[TL_MAP_CLASSES];   // first sorted position of every class
  int64_t gstart[TL_MAP_CLASSES];   // first group of every class
};

// rows_sorted = the rows by descending length (stable; classes are then contiguous): fills rowmap[groups * TL_RG] (pre-set to -1) and gload[groups] (zeroed)
template <typename I>
__global__ void __launch_bounds__(256) tl_map_build_kernel(int64_t M, int64_t cap, const I* __restrict__ indptr,
                                                           const int* __restrict__ rows_sorted, TlMapPlan plan,
                                                           int* __restrict__ rowmap, unsigned long long* __restrict__ gload) {
  GRID_STRIDE(i, M) {
    const int r = rows_sorted[i];
    const int64_t len = (int64_t)indptr[r + 1] - (int64_t)indptr[r];
    const int c = tl_map_class(len, cap);
    const int64_t j = i - plan.cstart[c];
    const int sl = tl_map_slots(c);
    const int64_t group = plan.gstart[c] + j / sl;
    rowmap[group * TL_RG + (int)(j % sl)] = r;
    if (len) atomicAdd(&gload[group], (unsigned long long)len);
  }
}

// One-pass inspector for CSR with sorted column indices and at most `direct_max_tiles` tiles: fills
// blk_off[groups * (tiles + 1)] (int32: per row group its tiles' first blocks and the end of its last list — pass
// SPAMD_TILED_GROUP_ENDS to spamd_spmm_tiled) and the block stream, which must have room for
// ceil(nnz / entries_per_block) + lists + slack blocks; state = one 64-bit word (zeroed here); on return state[0] != 0
// means a row with unsorted column indices was met: the outputs are then garbage and the caller takes the key-sort recipe.
extern "C" int spamd_spmm_tiled_inspect(int val_dtype, int idx_dtype, int64_t M, int64_t K, const void* a_data,
                                        const void* a_indices, const void* a_indptr, void* state, int* blk_off, int* blocks,
                                        void* stream) {
  if (M < 0 || K <= 0) return SPAMD_EINVAL;
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  const int64_t ntiles = ceil_div(K, (int64_t)TL_KB);
  if (ntiles > TL_DIRECT_MAX_TILES) return SPAMD_EINVAL;
  const int64_t groups = tl_grid_groups(M);
  hipError_t e = hipMemsetAsync(state, 0, sizeof(unsigned long long), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (groups == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    if (val_dtype == SPAMD_F32)
      return tl_launch_inspect<I, float>(M, ntiles, (const float*)a_data, (const I*)a_indices, (const I*)a_indptr,
                                         (unsigned long long*)state, blk_off, blocks, (hipStream_t)stream);
    return tl_launch_inspect<I, double>(M, ntiles, (const double*)a_data, (const I*)a_indices, (const I*)a_indptr,
                                        (unsigned long long*)state, blk_off, blocks, (hipStream_t)stream);
  })
  return SPAMD_ETYPE;
}

// Balanced layouts, step 1: keys[M] = K - length of every row (int64: the sort key - ascending = longest rows first; a row
// holds at most K elements), rows[M] = 0 .. M - 1 (the sort's payload), stats[8] (zeroed here): rows per class against `cap`
// [0..5], the largest natural group's stored elements [6].  The caller reads stats (one small read-back), decides (natural
// layout when [6] is close to the mean group), sorts rows by key (stable) and calls spamd_spmm_tiled_map_build.  keys = rows =
// NULL: the statistics only (what an operand that turns out NOT to be skewed pays: one pass over the row pointers).
extern "C" int spamd_spmm_tiled_map_stats(int idx_dtype, int64_t M, int64_t K, int64_t cap, const void* a_indptr, int64_t* keys,
                                          int* rows, int64_t* stats, void* stream) {
  if (M < 0 || M >= ((int64_t)1 << 31) || K <= 0 || cap < 32 || (cap & (cap - 1)) || (keys != nullptr) != (rows != nullptr) || !stats)
    return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(stats, 0, 8 * sizeof(int64_t), s); e != hipSuccess) return (int)e;
  if (M == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    hipLaunchKernelGGL((tl_map_skew_kernel<I>), dim3(tl_blocks_for(ceil_div(M, (int64_t)TL_RG))), dim3(256), 0, s, M, (const I*)a_indptr,
                       (unsigned long long*)stats);
    if (keys)
      hipLaunchKernelGGL((tl_map_stats_kernel<I>), dim3(tl_blocks_for(M)), dim3(256), 0, s, M, cap, K, (const I*)a_indptr, keys, rows,
                         (unsigned long long*)stats);
    return launch_status();
  })
  return SPAMD_ETYPE;
}

// groups of a balanced layout for these class counts (a multiple of the groups per workgroup); -1 on bad arguments
extern "C" int64_t spamd_spmm_tiled_map_groups(const int64_t* class_counts) {
  if (!class_counts) return -1;
  int64_t g = 0;
  for (int c = 0; c < TL_MAP_CLASSES; ++c) {
    if (class_counts[c] < 0) return -1;
    g += ceil_div(class_counts[c], (int64_t)tl_map_slots(c));
  }
  return ceil_div(g, (int64_t)TL_WAVES) * TL_WAVES;
}

// Balanced layouts, step 2: rows_sorted = the rows by descending length (stable sort of step 1's keys), class_counts = stats[0..5]
// (HOST array).  Fills rowmap[groups * rows_per_group + 16] (every unused slot -1) and gload[groups + 1] (stored elements of
// every group, then one zero: the caller's exclusive scan of it is the `vstart` of spamd_spmm_tiled_inspect_mapped).
extern "C" int spamd_spmm_tiled_map_build(int idx_dtype, int64_t M, int64_t cap, const void* a_indptr, const int* rows_sorted,
                                          const int64_t* class_counts, int* rowmap, int64_t* gload, void* stream) {
  const int64_t groups = spamd_spmm_tiled_map_groups(class_counts);
  if (M < 0 || M >= ((int64_t)1 << 31) || groups <= 0 || !rowmap || !gload || !rows_sorted) return SPAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(rowmap, 0xff, (size_t)(groups * TL_RG + 16) * sizeof(int), s); e != hipSuccess) return (int)e;
  if (hipError_t e = hipMemsetAsync(gload, 0, (size_t)(groups + 1) * sizeof(int64_t), s); e != hipSuccess) return (int)e;
  TlMapPlan plan;
  int64_t pos = 0, g = 0;
  for (int c = 0; c < TL_MAP_CLASSES; ++c) {
    plan.cstart[c] = pos;
    plan.gstart[c] = g;
    pos += class_counts[c];
    g += ceil_div(class_counts[c], (int64_t)tl_map_slots(c));
  }
  if (pos != M) return SPAMD_EINVAL;
  if (M == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    hipLaunchKernelGGL((tl_map_build_kernel<I>), dim3(tl_blocks_for(M)), dim3(256), 0, s, M, cap, (const I*)a_indptr, rows_sorted,
                       plan, rowmap, (unsigned long long*)gload);
    return launch_status();
  })
  return SPAMD_ETYPE;
}

// The one-pass inspector on a balanced layout: as spamd_spmm_tiled_inspect, with `groups` row groups whose rows come from
// `rowmap` and vstart[groups + 1] = stored elements before every group (and the total).  blk_off[groups * (tiles + 1)].
extern "C" int spamd_spmm_tiled_inspect_mapped(int val_dtype, int idx_dtype, int64_t M, int64_t K, int64_t groups,
                                               const void* a_data, const void* a_indices, const void* a_indptr,
                                               const int* rowmap, const int64_t* vstart, void* state, int* blk_off, int* blocks,
                                               void* stream) {
  if (M < 0 || K <= 0 || groups <= 0 || groups % TL_WAVES || !rowmap || !vstart) return SPAMD_EINVAL;
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  const int64_t ntiles = ceil_div(K, (int64_t)TL_KB);
  if (ntiles > TL_DIRECT_MAX_TILES) return SPAMD_EINVAL;
  hipError_t e = hipMemsetAsync(state, 0, sizeof(unsigned long long), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    if (val_dtype == SPAMD_F32)
      return tl_launch_inspect<I, float>(M, ntiles, (const float*)a_data, (const I*)a_indices, (const I*)a_indptr,
                                         (unsigned long long*)state, blk_off, blocks, (hipStream_t)stream, groups, rowmap, vstart);
    return tl_launch_inspect<I, double>(M, ntiles, (const double*)a_data, (const I*)a_indices, (const I*)a_indptr,
                                        (unsigned long long*)state, blk_off, blocks, (hipStream_t)stream, groups, rowmap, vstart);
  })
  return SPAMD_ETYPE;
}

// ---- the same block stream straight from CSC (late round 4) -------------------------------------------------------------
// The reference's default construction of a tall matrix is CSC (compressed_axes = argmin(shape)), and `a @ dense` then
// paid a CSC -> CSR sort of every stored element (4.3 ms of 7.4 at config 2's size) before the inspector above could run.
// But a K-tile's elements are CONTIGUOUS in CSC (160 whole columns), and inside a column the rows ascend, so a workgroup's
// 560 rows own one short run per column: no sort at all.
//   tl_csc_split_kernel   one pass over the row indices: split[c][bb] = first element (relative to the column's start) of
//                         column c whose row is >= 560 bb; also the "rows ascend inside every column" verdict.
//   tl_csc_count_kernel   workgroup = (560-row block = the 16 row groups of one executor workgroup, four tiles): the sizes of
//                         its (group, tile) lists, by walking its runs.
//   tl_csc_offsets_kernel blocks of every list, scanned per group; elements before every group (look-back over the
//                         workgroups, same launch) - groups are then placed by the SAME closed form as the CSR inspector.
//   tl_csc_fill_kernel    the same workgroups walk their runs again, consecutive lanes on consecutive elements; an element's
//                         place in its list comes from ballots over its wave + per-chunk counts in LDS.  Inside a list
//                         the entries are in column order (CSR inspector:
//                         row order) - an output element's terms stay k-ascending either way, which is all the products
//                         depend on.  (One workgroup per block walking all tiles one after the other - the first form -
//                         is a chain of ~130 dependent steps per workgroup: 2.1 / 2.6 ms.)
constexpr int TL_BLOCK_ROWS = TL_RG * TL_WAVES;

template <typename I>
__global__ void __launch_bounds__(256) tl_csc_split_kernel(int64_t M, int64_t K, int64_t nblocks, const I* __restrict__ indices,
                                                           const I* __restrict__ indptr, int* __restrict__ split,
                                                           unsigned long long* __restrict__ state,
                                                           unsigned long long* __restrict__ look, int nlook) {
  const int tid = threadIdx.x;
  bool bad = false;
  if (blockIdx.x == 0)
    for (int i = tid; i < nlook; i += 256) look[i] = 0;       // (tl_csc_offsets_kernel's ticket and state words)
  for (int64_t c = blockIdx.x; c < K; c += gridDim.x) {
    const int64_t a = (int64_t)indptr[c], b = (int64_t)indptr[c + 1];
    if (b - a >= ((int64_t)1 << 31)) bad = true;
    if (a >= b) {
      for (int64_t bb = tid; bb <= nblocks; bb += 256) split[c * (nblocks + 1) + bb] = 0;
      continue;
    }
#pragma unroll 4
    for (int64_t e = a + tid; e < b; e += 256) {
      const int64_t r = (int64_t)indices[e];
      const int64_t rp = e > a ? (int64_t)indices[e - 1] : -1;
      if (rp > r || r < 0 || r >= M) bad = true;
      int64_t bc = r / TL_BLOCK_ROWS, bp = e > a ? rp / TL_BLOCK_ROWS : -1;
      if (bc < 0) bc = 0;
      if (bc >= nblocks) bc = nblocks - 1;
      if (bp >= nblocks) bp = nblocks - 1;
      for (int64_t bb = bp + 1; bb <= bc; ++bb) split[c * (nblocks + 1) + bb] = (int)(e - a);
      if (e == b - 1)
        for (int64_t bb = bc + 1; bb <= nblocks; ++bb) split[c * (nblocks + 1) + bb] = (int)(b - a);
    }
  }
  if (bad) atomicOr(&state[0], 1ull);
}

#ifndef SPAMD_CSC_ABL
#define SPAMD_CSC_ABL 0   // timing ablations of tl_csc_fill_kernel (wrong streams): 1 no stores to the stream, 2 no ranks, 3 no value
                          // loads, 4 no element loads at all, 5 no image (neither written nor copied out), 6 the runs of every tile only
#endif
constexpr int TL_CSC_IMG_BLOCKS = 256;   // 64-byte blocks of a tile's 16 lists assembled in LDS (~190 for float64 at 1 %)
constexpr int TL_CSC_STAGE = 1024;   // elements of a (row block, tile) staged in LDS at a time (a tile's runs hold ~900 at 1 %)
#ifndef SPAMD_CSC_TC
#define SPAMD_CSC_TC 4
#endif
constexpr int TL_CSC_TC = SPAMD_CSC_TC;         // tiles per workgroup (count and fill kernels: grid = row blocks x tile chunks)

// the runs of tile t of row block b: rstart[j] = first element of the block's run in column j of the tile, pre[j] =
// exclusive prefix of the run lengths (pre[TL_KB] = the tile's element count, returned to every thread).  The tile's
// elements in (column, row) order are then positions 0 .. total - 1; tl_csc_locate maps a position to its element, so that
// consecutive lanes read consecutive elements of a run (a thread per column instead - the first form of these kernels -
// costs the texture addresser one cache line per LANE: 3.6 / 5.2 ms at config 2's size against 2.1 / 2.6 ms).
// (the prefix part: `len` = the length of thread tid's run, 0 for the threads past the tile's columns)
__device__ __forceinline__ int tl_csc_scan_runs(int len, int* pre, int* wsum) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int x = (int)wave_incl_scan_u32((unsigned)len);       // (DPP: no LDS round trips)
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  int off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    off += w < wv ? wsum[w] : 0;
    total += wsum[w];
  }
  if (tid < TL_KB) pre[tid] = off + x - len;
  if (tid == 0) pre[TL_KB] = total;
  __syncthreads();
  return total;
}

template <typename I>
__device__ __forceinline__ int tl_csc_runs(int t, int64_t K, const I* __restrict__ indptr, const int* __restrict__ sp,
                                           int64_t pitch, long long* rstart, int* pre, int* wsum) {
  const int tid = threadIdx.x;
  const int64_t c = (int64_t)t * TL_KB + tid;
  int len = 0;
  if (tid < TL_KB && c < K) {
    const int a0 = sp[c * pitch], a1 = sp[c * pitch + 1];   // (this block's and the next one's pointer: adjacent words)
    len = a1 - a0;
    rstart[tid] = (int64_t)indptr[c] + a0;
  }
  return tl_csc_scan_runs(len, pre, wsum);
}

__device__ __forceinline__ int64_t tl_csc_locate(int k, const long long* rstart, const int* pre) {
  int lo = 0, hi = TL_KB - 1;       // largest j with pre[j] <= k (pre[0] = 0; empty columns are skipped: they share a prefix)
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    const int mid = (lo + hi + 1) >> 1;
    const bool le = pre[mid] <= k;
    lo = le ? mid : lo;
    hi = le ? hi : mid - 1;
  }
  return rstart[lo] + (k - pre[lo]);
}

// cnt[t * groups + g] = elements of list (g, t)
template <typename I>
__global__ void __launch_bounds__(256) tl_csc_count_kernel(int64_t K, int ntiles, int64_t nblocks, int64_t groups,
                                                           const I* __restrict__ indices,
                                                           const I* __restrict__ indptr, const int* __restrict__ split,
                                                           const unsigned long long* __restrict__ state, int* __restrict__ cnt) {
  __shared__ long long rstart[TL_KB];
  __shared__ int pre[TL_KB + 1];
  __shared__ int wsum[4];
  __shared__ int c16[TL_WAVES];
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t r_base = b * TL_BLOCK_ROWS;
  if (__hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  const int t_end = ((int)blockIdx.y + 1) * TL_CSC_TC < ntiles ? ((int)blockIdx.y + 1) * TL_CSC_TC : ntiles;
  for (int t = (int)blockIdx.y * TL_CSC_TC; t < t_end; ++t) {
    if (tid < TL_WAVES) c16[tid] = 0;
    const int total = tl_csc_runs<I>(t, K, indptr, split + b, nblocks + 1, rstart, pre, wsum);
    for (int k = tid; k < total; k += 256) {
      const int64_t e = tl_csc_locate(k, rstart, pre);
      atomicAdd(&c16[(int)(((int64_t)indices[e] - r_base) / TL_RG)], 1);
    }
    __syncthreads();
    if (tid < TL_WAVES) cnt[(int64_t)t * groups + b * TL_WAVES + tid] = c16[tid];
    __syncthreads();
  }
}

// The same counts without the runs (late round 4): a workgroup takes a QUARTER of a tile's columns (whole columns: rows of
// every block) and histograms the row groups of their elements in LDS (4 bytes per group: up to TL_CSC_HIST_GROUPS groups);
// its counts go to its own quarter of `cntq`, which tl_csc_offsets_kernel adds up.  No position search, every row index
// read once, coalesced: 0.41 / 0.45 ms (count kernel above, int32 / int64 indices at config 2's size) -> 0.26 / 0.31 ms; the
// split pointers of its columns are written in the same pass (tl_csc_split_kernel is then not launched: 0.28 / 0.34 ms).
// (round 6: two groups share a 32-bit word - a part of a tile holds at most TL_RG * TL_KB / parts = 1400 elements of a group, so
// 16 bits per count and one 32-bit LDS atomic of 1 or 65536 - which takes the histogram to 76 k groups = 2.66 M rows; above
// that the split + count kernels: +0.48 ms at 4 M x 2500)
constexpr int TL_CSC_HIST_GROUPS = 76 * 1024;   // 152 KB of LDS
#ifndef SPAMD_CSC_HIST_PARTS
#define SPAMD_CSC_HIST_PARTS 4
#endif
constexpr int TL_CSC_HIST_PARTS = SPAMD_CSC_HIST_PARTS;

template <typename I>
__global__ void __launch_bounds__(1024) tl_csc_hist_kernel(int64_t M, int64_t K, int ntiles, int64_t groups, int64_t nblocks,
                                                           const I* __restrict__ indices, const I* __restrict__ indptr,
                                                           int* __restrict__ cntq, int* __restrict__ split,
                                                           unsigned long long* __restrict__ state,
                                                           unsigned long long* __restrict__ look, int nlook) {
  extern __shared__ int tl_hist_lds[];
  const int tid = threadIdx.x;
  const int t = blockIdx.x, q = blockIdx.y;
  if (t == 0 && q == 0)
    for (int i = tid; i < nlook; i += 1024) look[i] = 0;       // (tl_csc_offsets_kernel's ticket and state words)
  constexpr int CPQ = TL_KB / TL_CSC_HIST_PARTS;
  constexpr int U = 4;                         // loads in flight per thread (x 2: the row and the row in front)
  __shared__ long long cptr[CPQ + 1];          // the pointers of my columns
  int64_t c0 = (int64_t)t * TL_KB + q * CPQ, c1 = c0 + CPQ;
  if (c0 > K) c0 = K;
  if (c1 > K) c1 = K;
  const int ncols = (int)(c1 - c0);
  if (tid <= ncols) cptr[tid] = (int64_t)indptr[c0 + tid];
  static_assert(TL_RG * (TL_KB / TL_CSC_HIST_PARTS) < 65536, "a group's count in a part of a tile fits 16 bits");
  const int64_t hwords = (groups + 1) / 2;
  for (int64_t i = tid; i < hwords; i += 1024) tl_hist_lds[i] = 0;
  __syncthreads();
  // The split pointers of my columns in the same pass (tl_csc_split_kernel's work).  Round 6: my columns' elements are ONE
  // contiguous range of the arrays; the workgroup walks it 4096 elements at a time with every load of a step issued before
  // the first is used (a column at a time, one load per thread and step, a wave went through ~400 memory round trips one
  // behind the other), every thread keeps the column of its position by stepping a cursor through the pointers, and the
  // arithmetic per element is 32-bit: positions relative to the step's first element (the loads take a uniform base and a
  // 32-bit offset), rows as unsigned words (this kernel serves at most TL_CSC_HIST_GROUPS * TL_RG rows; a row outside
  // [0, M) is reported by its own element, so the row in front is read as its low word only).  With 64-bit positions
  // and three 64-bit divisions by constants per element the kernel was bound by its own instructions: 0.40 ms at config 2's
  // size (int64 indices) for 0.8 GB read.
  bool bad = false;
  const unsigned nb1 = (unsigned)nblocks - 1u, ng1 = (unsigned)groups - 1u, Mu = (unsigned)M;
  for (int j = 0; j < ncols; ++j) {
    const int64_t a = cptr[j], b = cptr[j + 1];
    if (b - a >= ((int64_t)1 << 31)) bad = true;
    if (a >= b)
      for (int64_t bb = tid; bb <= nblocks; bb += 1024) split[(c0 + j) * (nblocks + 1) + bb] = 0;
  }
  const int64_t A = cptr[0], B = cptr[ncols];
  int cj = 0;
  int64_t ca = A, cb = cptr[ncols > 0 ? 1 : 0];
  int* sp = split + c0 * (nblocks + 1);
  auto rel = [](int64_t d) { return (int)(d < -0x7fffffffll ? -0x7fffffffll : d > 0x7fffffffll ? 0x7fffffffll : d); };
  // (the loads of a step are issued one step ahead, from clamped places past the end: a wave's memory round trip and its
  // work on the step before overlap)
  unsigned nlo[U], nhi[U], npl[U];
  auto load = [&](int64_t base) {
    const int64_t left = B - base;
    const int last = left <= 0 ? 0 : (int)(left < 1024 * U ? left : 1024 * U) - 1;
    const int64_t from = left <= 0 ? B - 1 : base;
    const int shift = from > 0 ? 0 : 1;                 // (the element in front of the very first one does not exist)
    const I* const pb = indices + from;
    const I* const pbm = pb - 1 + shift;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int o = u * 1024 + tid, oc = o < last ? o : last;
      if constexpr (sizeof(I) == 8) {
        typedef unsigned uint2v __attribute__((ext_vector_type(2)));
        const uint2v w = *reinterpret_cast<const uint2v*>(pb + oc);
        nlo[u] = w.x;
        nhi[u] = w.y;
      } else {
        nlo[u] = (unsigned)pb[oc];
        nhi[u] = 0;       // (a negative row is a large unsigned one)
      }
      npl[u] = *reinterpret_cast<const unsigned*>(pbm + ((oc > shift ? oc : shift) - shift));
    }
  };
  if (B > A) load(A);
  for (int64_t base = A; base < B; base += 1024 * U) {
    const int nrem = (int)(B - base < 1024 * U ? B - base : 1024 * U);
    unsigned rlo[U], rhi[U], plo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rlo[u] = nlo[u];
      rhi[u] = nhi[u];
      plo[u] = npl[u];
    }
    load(base + 1024 * U);
    int rel_a = rel(ca - base), rel_b = rel(cb - base);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int o = u * 1024 + tid;
      if (o < nrem) {
        while (o >= rel_b && cj + 1 < ncols) {       // (empty columns are stepped over)
          ++cj;
          ca = cb;
          cb = cptr[cj + 1];
          sp += nblocks + 1;
          rel_a = rel(ca - base);
          rel_b = rel(cb - base);
        }
        const bool first = o == rel_a;
        if (rhi[u] != 0 || rlo[u] >= Mu || (!first && plo[u] > rlo[u])) bad = true;
        unsigned g = rlo[u] / (unsigned)TL_RG;       // (rows out of range are reported above; stay inside the histogram)
        g = g < ng1 ? g : ng1;
        atomicAdd(&tl_hist_lds[g >> 1], (g & 1u) ? 65536 : 1);
        const int bc = (int)(g / (unsigned)TL_WAVES);       // (= row / TL_BLOCK_ROWS, at most nblocks - 1)
        int bp = -1;
        if (!first) {
          const unsigned gp = plo[u] / (unsigned)TL_BLOCK_ROWS;
          bp = (int)(gp < nb1 ? gp : nb1);
        }
        for (int bb = bp + 1; bb <= bc; ++bb) sp[bb] = o - rel_a;
        if (o == rel_b - 1)
          for (int64_t bb = (int64_t)bc + 1; bb <= nblocks; ++bb) sp[bb] = rel_b - rel_a;
      }
    }
  }
  __shared__ long long counted;
  if (tid == 0) counted = 0;
  if (bad) atomicOr(&state[0], 1ull);
  __syncthreads();
  int* const out = cntq + (int64_t)q * groups * ntiles;
  long long mine = 0;
  for (int64_t g = tid; g < groups; g += 1024) {       // (tile-major: written and read coalesced)
    const int c = (int)(((unsigned)tl_hist_lds[g >> 1] >> (16 * (int)(g & 1))) & 0xffffu);
    out[(int64_t)t * groups + g] = c;
    mine += c;
  }
  // A 16-bit count that wrapped (only repeated rows inside a column can do that: 65536 elements of one row group in 40
  // columns) changes the sum of the counts: reported like rows out of order - the caller takes the CSR route.
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
  if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counted), (unsigned long long)mine);
  __syncthreads();
  if (tid == 0 && counted != B - A) atomicOr(&state[0], 1ull);
}

// per group: its lists' first blocks relative to the group's own (rel[t * groups + g], tile-major like cnt; row ntiles = the group's
// blocks) and e0[g] = the elements of the groups before g (e0[groups] = all of them).  The scan over the groups runs in the
// same launch (round 6): workgroups take tickets and look back over one state word per workgroup (common.h); `look` =
// ticket word, a spare word, then the state words - zeroed by the kernel in front (histogram or split kernel).  As a
// launch of its own (one workgroup, every thread its own stretch of the groups: 28 dependent loads) the scan took 50 us at
// config 2's size.
// One WAVE per workgroup and eight tiles' counts loaded at a time: a thread per group walking its 63 x 4 counts one load
// behind the other, 112 workgroups of 256 threads, took 47 us at config 2's size (29 MB read).
constexpr int TL_CSC_OFF_TB = 8;
template <int EPB, int PARTS>
__global__ void __launch_bounds__(64) tl_csc_offsets_kernel(int64_t groups, int ntiles, const int* __restrict__ cnt,
                                                            const unsigned long long* __restrict__ state,
                                                            int* __restrict__ rel, unsigned long long* __restrict__ look,
                                                            long long* __restrict__ e0) {
  constexpr int TB = TL_CSC_OFF_TB;
  const int lane = threadIdx.x;
  long long ticket = 0;
  if (lane == 0) ticket = (long long)atomicAdd(look, 1ull);
  const int64_t blk = __shfl(ticket, 0, 64);
  const int64_t g = blk * 64 + lane;
  const int64_t gc = g < groups ? g : groups - 1;
  const bool bad = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  long long tot = 0;
  int run = 0;
  for (int t0 = 0; t0 < ntiles; t0 += TB) {
    int c[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int t = t0 + j < ntiles ? t0 + j : ntiles - 1;
      c[j] = 0;
#pragma unroll
      for (int q = 0; q < PARTS; ++q) c[j] += cnt[((int64_t)q * ntiles + t) * groups + gc];
    }
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      if (t0 + j < ntiles && g < groups) {
        const int cc = bad ? 0 : c[j];
        rel[(int64_t)(t0 + j) * groups + g] = run;
        run += (cc + EPB - 1) / EPB;
        tot += cc;
      }
    }
  }
  if (g < groups) rel[(int64_t)ntiles * groups + g] = run;
  long long incl = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  const long long total = __shfl(incl, 63, 64);
  const long long base = (long long)lookback_exclusive(look + 2, blk, (unsigned long long)total, lane);
  if (g < groups) e0[g] = base + incl - tot;
  if (g == groups - 1) e0[groups] = base + incl;
}

template <typename I, typename T>
__global__ void __launch_bounds__(256) tl_csc_fill_kernel(int64_t K, int ntiles, const T* __restrict__ vals,
                                                          const I* __restrict__ indices, const I* __restrict__ indptr,
                                                          const int* __restrict__ split,
                                                          const unsigned long long* __restrict__ state,
                                                          const int* __restrict__ rel, const long long* __restrict__ e0,
                                                          int64_t nblocks, int* __restrict__ blk_off, int* __restrict__ stream) {
  constexpr int EPB = TlFmt<T>::EPB;
  constexpr int GPB = TL_WAVES;               // row groups of a block
  constexpr int PER = TL_CSC_STAGE / 256;     // elements of a window per thread
  constexpr int CHUNKS = TL_CSC_STAGE / 64;   // 64-element chunks of a window (one per wave and step)
  __shared__ long long rstart[TL_KB];
  __shared__ int pre[TL_KB + 1];
  __shared__ int wsum[4];
  __shared__ long long goff_s[GPB];
  __shared__ int tbase[GPB], lo16[GPB];
  __shared__ int ccnt[CHUNKS * GPB];          // elements of a chunk per group, then their exclusive prefix over the window
  __shared__ __attribute__((aligned(16))) int img[TL_CSC_IMG_BLOCKS * TL_BLOCK_INTS];   // the tile's lists as they go to the stream
  __shared__ int ib[GPB], tsum_s;
  __shared__ int gbase[GPB];                  // first block of the tile's list of every group (what blk_off holds)
  __shared__ int bdst[TL_CSC_IMG_BLOCKS];     // where every block of the LDS image goes (written by the block's entries)
  __shared__ unsigned char colof[TL_CSC_STAGE];   // column (inside the tile) of every position of the current window
  static_assert(TL_KB <= 256, "a tile's columns fit a byte");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // workgroup L runs on XCD L % 8 (observed placement; for speed only): an XCD takes a contiguous eighth of the row blocks,
  // consecutive workgroups of an XCD = consecutive blocks of the same tiles - their runs share cache lines (a 128-byte line
  // of row indices spans 3-6 blocks), which are then found in that XCD's L2
  const int64_t groups = nblocks * GPB;
  const int64_t per_x = (nblocks + 7) / 8;
  const int64_t seq = (int64_t)(blockIdx.x >> 3);
  const int64_t b = (int64_t)(blockIdx.x & 7u) * per_x + seq % per_x;
  if (b >= nblocks) return;
  const int64_t r_base = b * TL_BLOCK_ROWS;
  const int t_beg = (int)(seq / per_x) * TL_CSC_TC;
  const int t_end = t_beg + TL_CSC_TC < ntiles ? t_beg + TL_CSC_TC : ntiles;
  const bool last_chunk = t_end == ntiles;
  if (__hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
    // rows that do not ascend inside a column: the runs are meaningless.  Empty lists (memory-safe for the one product that
    // runs before the caller reads the verdict and takes the CSR route).
    for (int i = tid; i < GPB * (t_end - t_beg + (last_chunk ? 1 : 0)); i += 256) {
      const int gi = i % GPB, t = t_beg + i / GPB;
      blk_off[(b * GPB + gi) * (ntiles + 1) + t] = 0;
    }
    return;
  }
  if (tid < GPB) {
    const int64_t g = b * GPB + tid;
    goff_s[tid] = tl_group_first_block(e0[g], g, ntiles, EPB);
  }
  __syncthreads();
  if (last_chunk) {
    // the group's end and the zeroed gap in front of the next group
    for (int gi = 0; gi < GPB; ++gi) {
      const int64_t g = b * GPB + gi;
      const int64_t gend = goff_s[gi] + rel[(int64_t)ntiles * groups + g];
      const int64_t gnext = tl_group_first_block(e0[g + 1], g + 1, ntiles, EPB);
      if (tid == 0) blk_off[g * (ntiles + 1) + ntiles] = (int)gend;
      for (int64_t i = gend * TL_BLOCK_INTS + tid; i < gnext * TL_BLOCK_INTS; i += 256) stream[i] = 0;
    }
  }
  const unsigned long long below = (1ull << lane) - 1;
  // A tile's runs (split pointers, column pointer) and its lists' offsets are loaded one tile ahead, by every thread from
  // clamped places: three memory round trips one behind the other per tile otherwise (offsets, runs, elements - round 6).
  // The loads are issued behind the element loads of the tile before and taken over (`take`: real moves, so that the wait
  // for them stands there) in front of that tile's stores to the stream: at the head of the loop the wait would cover the
  // stores as well.
  int nx_a0, nx_a1, nx_rel, cur_a0, cur_a1, cur_rel;
  long long nx_ip, cur_ip;
  auto fetch = [&](int t) {
    int64_t c = (int64_t)t * TL_KB + tid;
    if (tid >= TL_KB || c >= K) c = (int64_t)t * TL_KB;
    const int* const q = split + b + c * (nblocks + 1);   // (this block's and the next one's pointer: adjacent words)
    nx_a0 = q[0];
    nx_a1 = q[1];
    nx_ip = (long long)indptr[c];
    nx_rel = rel[(int64_t)t * groups + b * GPB + (tid & (GPB - 1))];
  };
  auto take = [&]() {
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b64 %3, %7"
                 : "=&v"(cur_a0), "=&v"(cur_a1), "=&v"(cur_rel), "=&v"(cur_ip)
                 : "v"(nx_a0), "v"(nx_a1), "v"(nx_rel), "v"(nx_ip));
  };
  fetch(t_beg);
  take();
  for (int t = t_beg; t < t_end; ++t) {
    const int a0 = cur_a0, a1 = cur_a1, rl = cur_rel;
    const long long ip = cur_ip;
    const int t_next = t + 1 < t_end ? t + 1 : t;
    if (tid < GPB) {
      tbase[tid] = 0;
      lo16[tid] = rl;
    }
    int len = 0;
    if (tid < TL_KB && (int64_t)t * TL_KB + tid < K) {
      len = a1 - a0;
      rstart[tid] = ip + a0;
    }
    const int total = tl_csc_scan_runs(len, pre, wsum);
#if SPAMD_CSC_ABL == 6
    fetch(t_next);
    take();
    continue;
#endif
    if (total <= 0) {       // (no window below)
      fetch(t_next);
      take();
    }
    bool padded = false;
    if (tid < GPB) {
      gbase[tid] = (int)(goff_s[tid] + lo16[tid]);
      blk_off[(b * GPB + tid) * (ntiles + 1) + t] = gbase[tid];
    }
    // The tile's elements in (column, row) order, a window at a time: consecutive lanes take consecutive elements (of a
    // run), an element's place inside its list = the elements of its group in front of it: its rank among the equal-group
    // lanes of its wave (ballots) + the group's count in the chunks before (LDS).  Lanes of one group write consecutive
    // entries: a store instruction touches ~16 runs of lines instead of 64 lines.
    for (int w0 = 0; w0 < total; w0 += TL_CSC_STAGE) {
      const int w1 = w0 + TL_CSC_STAGE < total ? w0 + TL_CSC_STAGE : total;
      int rr[PER], rank[PER], col[PER], grp[PER];
      T vv[PER];
      for (int i = tid; i < CHUNKS * GPB; i += 256) ccnt[i] = 0;
      // position -> column of the window, written by the columns themselves (a run holds ~6 positions): an element then finds
      // its column with ONE LDS read instead of an 8-step binary search over the runs' prefix - eight DEPENDENT LDS round
      // trips per element on the critical path of every window (round 5)
      if (tid < TL_KB) {
        const int a = pre[tid] > w0 ? pre[tid] : w0, b = pre[tid + 1] < w1 ? pre[tid + 1] : w1;
        for (int k = a; k < b; ++k) colof[k - w0] = (unsigned char)tid;
      }
      __syncthreads();
      // every load of the window first, from clamped positions (a load under `if (valid)` is waited for on the spot: four
      // memory round trips one behind the other per window - round 6), then the ranks
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        const int k = w0 + tid + 256 * p;
        const int kc = k < w1 ? k : w1 - 1;
        const int lo = colof[kc - w0];       // the column whose run holds position k
        const int64_t e = rstart[lo] + (kc - pre[lo]);
        col[p] = lo;
        // (the low word of the row: its distance from the block's first row fits an int)
#if SPAMD_CSC_ABL == 4
        rr[p] = (int)((unsigned)(e * 7 + p) % (unsigned)TL_BLOCK_ROWS);
#else
        rr[p] = (int)(*reinterpret_cast<const unsigned*>(indices + e) - (unsigned)r_base);
#endif
#if SPAMD_CSC_ABL == 3 || SPAMD_CSC_ABL == 4
        vv[p] = T(1);
#else
        vv[p] = vals[e];
#endif
      }
      {       // (once more per further window of a long tile: harmless.  The tile number goes through an empty asm so that
              // the loads stay HERE, behind the element loads: hoisted to the head of the window they are waited for first)
        int tn = t_next;
        asm volatile("" : "+s"(tn));
        fetch(tn);
      }
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        const int k = w0 + tid + 256 * p;
        rank[p] = 0;
        grp[p] = 0;
        if (k < w1) {       // (the ballots see the valid lanes only)
          const int gi = rr[p] / TL_RG;
          grp[p] = gi;
          // the lanes of my group: per bit of the group number, the lanes that agree with me (bit set: the ballot, clear: its
          // complement) - the bit as 0 / all ones (v_bfe_i32, opaque to the compiler: its own choice was two compares and a
          // select per bit), one XNOR + AND per half of the mask
          const unsigned long long act = __ballot(1);
          unsigned mlo = (unsigned)act, mhi = (unsigned)(act >> 32);
#if SPAMD_CSC_ABL == 2
          mlo = lane < 32 ? 1u << lane : 0u;
          mhi = lane < 32 ? 0u : 1u << (lane - 32);
#else
#pragma unroll
          for (int bit = 0; bit < 4; ++bit) {
            int x;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(x) : "v"(gi), "n"(bit));
            const unsigned long long bb = __ballot(x != 0);
            mlo &= ~((unsigned)bb ^ (unsigned)x);
            mhi &= ~((unsigned)(bb >> 32) ^ (unsigned)x);
          }
#endif
          rank[p] = __popc(mlo & (unsigned)below) + __popc(mhi & (unsigned)(below >> 32));
          if (rank[p] == 0) ccnt[(wv + 4 * p) * GPB + gi] = __popc(mlo) + __popc(mhi);
        }
      }
      __syncthreads();
      {  // exclusive prefix over the window's chunks, per group (thread = (chunk, group)); the tile's running totals
        static_assert(CHUNKS * GPB == 256, "one thread per (chunk, group)");
        const int gi = tid & (GPB - 1), ch = tid / GPB;
        int sum = tbase[gi], all = 0;
        for (int c = 0; c < CHUNKS; ++c) {
          const int x = ccnt[c * GPB + gi];
          sum += c < ch ? x : 0;
          all += x;
        }
        __syncthreads();
        ccnt[ch * GPB + gi] = sum;
        if (ch == 0) {
          tbase[gi] += all;
          // blocks of the tile's lists and their starts inside the LDS image (meaningful when the tile is one window)
          const int nb = (tbase[gi] + EPB - 1) / EPB;
          static_assert(GPB == 16, "a DPP row");
          unsigned x = (unsigned)nb;       // (inclusive scan over the 16 lanes of a DPP row)
          x += dpp_shift0<0x111, 0xf>(x);
          x += dpp_shift0<0x112, 0xf>(x);
          x += dpp_shift0<0x114, 0xf>(x);
          x += dpp_shift0<0x118, 0xf>(x);
          ib[gi] = (int)x - nb;
          if (gi == GPB - 1) tsum_s = x;
        }
      }
      __syncthreads();
      // A tile in one window whose lists fit the LDS image (the usual case) is assembled there - padding included - and
      // copied out with consecutive lanes on consecutive 16 bytes of a list: an entry's own stores (8 bytes; float64: 4 + 8 in
      // two places of its block) cost this kernel 0.3 ms (float32) / 0.6 ms (float64) of 1.0 / 1.3 at config 2's size.
      const bool img_path = total <= TL_CSC_STAGE && tsum_s <= TL_CSC_IMG_BLOCKS;     // (tsum_s: written with the prefix above)
      if (img_path) {
        for (int i = tid; i < tsum_s * TL_BLOCK_INTS; i += 256) img[i] = 0;
        __syncthreads();
      }
#if SPAMD_CSC_ABL == 5
      if (false)
#endif
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        const int k = w0 + tid + 256 * p;
        if (k < w1) {
          const int gi = grp[p], lr = rr[p] - gi * TL_RG;
          const int pos = ccnt[(wv + 4 * p) * GPB + gi] + rank[p];
          if (img_path) {
            const int q = pos / EPB, slot = pos - q * EPB;       // block of the list, entry of the block
            const int base = (ib[gi] + q) * TL_BLOCK_INTS;
            const int d0 = tl_d0(col[p], lr);
            // the block's place in the stream, by every entry of the block (a block of the image has at least one): the
            // copy below then reads one word instead of searching the lists' starts - 15 compares per 16 bytes copied (round 6)
            bdst[ib[gi] + q] = gbase[gi] + q;
            if constexpr (sizeof(T) == 4) {
              typedef int int2v __attribute__((ext_vector_type(2)));
              int2v ent;
              ent.x = d0;
              ent.y = __builtin_bit_cast(int, vv[p]);
              *reinterpret_cast<int2v*>(&img[base + slot * 2]) = ent;
            } else {
              img[base + slot] = d0;
              *reinterpret_cast<long long*>(&img[base + 6 + 2 * slot]) = __builtin_bit_cast(long long, vv[p]);
            }
          } else {
            const int64_t dst = (int64_t)gbase[gi] * EPB + pos;
            TlFmt<T>::put(stream, dst, tl_d0(col[p], lr), vv[p]);
          }
        }
      }
      __syncthreads();
      // every element load counts as used on every path from here (a window without valid lanes in some wave skips the
      // uses above, and the loads would reach the next window's first register write "still in flight": a full wait there)
#pragma unroll
      for (int p = 0; p < PER; ++p) asm volatile("" ::"v"(vv[p]), "v"(rr[p]));
      take();
      if (img_path && SPAMD_CSC_ABL != 5) {
        typedef int int4v __attribute__((ext_vector_type(4)));
        const int nvec = tsum_s * (TL_BLOCK_INTS / 4);
        for (int i = tid; i < nvec; i += 256) {
          const int64_t dst = (int64_t)bdst[i >> 2] * TL_BLOCK_INTS + (i & 3) * 4;
#if SPAMD_CSC_ABL == 1
          if (bdst[i >> 2] == 0x7fffffff)
#endif
          *reinterpret_cast<int4v*>(stream + dst) = *reinterpret_cast<const int4v*>(&img[i * 4]);
        }
        __syncthreads();
        padded = true;
      }
    }
    // padding entries of the tile's lists (zero d0 and value: they accumulate into the junk register pair); the LDS image
    // carried them already
    if (!padded && tid < GPB) {
      const int c = tbase[tid], nb = (c + EPB - 1) / EPB;
      const int64_t first = (goff_s[tid] + lo16[tid]) * EPB;
      for (int q = c; q < nb * EPB; ++q) TlFmt<T>::put(stream, first + q, 0, T(0));
    }
    __syncthreads();
  }
}

template <typename I, typename T>
static int tl_launch_inspect_csc(int64_t M, int64_t K, int64_t ntiles, const T* a_data, const I* a_indices, const I* a_indptr,
                                 int* ws, unsigned long long* state, int* blk_off, int* blocks, hipStream_t s) {
  const int64_t groups = tl_grid_groups(M);
  const int64_t nblocks = groups / TL_WAVES;
  if (nblocks >= ((int64_t)1 << 31)) return SPAMD_EINVAL;
  // workspace: split[K * (nblocks + 1)] (column-major: a column's pointers are written next to each other - block-major, every
  // 4-byte pointer dirtied a cache line of its own: 0.8 GB written back for 0.07 GB of pointers), cnt[groups * ntiles], rel[groups * (ntiles + 1)], look[groups] (ticket + state words of the offsets kernel), e0[groups + 1] (8-byte)
  const bool hist = groups <= TL_CSC_HIST_GROUPS;
  int* const split = ws;
  int* const cnt = split + (nblocks + 1) * K;
  int* const rel = cnt + TL_CSC_HIST_PARTS * groups * ntiles;
  const int64_t words = ((nblocks + 1) * K + TL_CSC_HIST_PARTS * groups * ntiles + groups * (ntiles + 1) + 1) & ~(int64_t)1;   // (8-byte alignment)
  // (`groups` 8-byte words - the per-group counts of the first form - hold the look-back words of the offsets kernel)
  unsigned long long* const look = reinterpret_cast<unsigned long long*>(ws + words);
  long long* const e0 = reinterpret_cast<long long*>(look) + groups;
  const int64_t owgs = ceil_div(groups, (int64_t)64);
  const int nlook = (int)(owgs + 2);
  static_assert(TL_WAVES >= 3, "groups (a multiple of TL_WAVES) >= ceil(groups / 64) + 2");
  if (!hist) {
    const unsigned sgrid = (unsigned)std::min<int64_t>(K, (int64_t)256 * 64);
    hipLaunchKernelGGL((tl_csc_split_kernel<I>), dim3(sgrid), dim3(256), 0, s, M, K, nblocks, a_indices, a_indptr, split, state, look, nlook);
  }
  const dim3 grid((unsigned)nblocks, (unsigned)ceil_div(ntiles, (int64_t)TL_CSC_TC));
  if (hist) {
    auto hk = &tl_csc_hist_kernel<I>;
    const int lds = (int)((groups + 1) / 2 * sizeof(int));
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hk), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(hk, dim3((unsigned)ntiles, TL_CSC_HIST_PARTS), dim3(1024), lds, s, M, K, (int)ntiles, groups, nblocks,
                       a_indices, a_indptr, cnt, split, state, look, nlook);
  } else {
    hipLaunchKernelGGL((tl_csc_count_kernel<I>), grid, dim3(256), 0, s, K, (int)ntiles, nblocks, groups, a_indices, a_indptr, (const int*)split,
                       (const unsigned long long*)state, cnt);
  }
  if (hist)
    hipLaunchKernelGGL((tl_csc_offsets_kernel<TlFmt<T>::EPB, TL_CSC_HIST_PARTS>), dim3((unsigned)owgs), dim3(64), 0, s,
                       groups, (int)ntiles, (const int*)cnt, (const unsigned long long*)state, rel, look, e0);
  else
    hipLaunchKernelGGL((tl_csc_offsets_kernel<TlFmt<T>::EPB, 1>), dim3((unsigned)owgs), dim3(64), 0, s,
                       groups, (int)ntiles, (const int*)cnt, (const unsigned long long*)state, rel, look, e0);
  const int64_t fgrid = 8 * ceil_div(nblocks, (int64_t)8) * ceil_div(ntiles, (int64_t)TL_CSC_TC);
  if (fgrid >= ((int64_t)1 << 31)) return SPAMD_EINVAL;
  hipLaunchKernelGGL((tl_csc_fill_kernel<I, T>), dim3((unsigned)fgrid), dim3(256), 0, s, K, (int)ntiles, a_data, a_indices,
                     a_indptr, (const int*)split, (const unsigned long long*)state, (const int*)rel, (const long long*)e0, nblocks,
                     blk_off, blocks);
  return launch_status();
}

// int32 words of workspace spamd_spmm_tiled_inspect_csc needs
extern "C" int64_t spamd_spmm_tiled_inspect_csc_ws(int64_t M, int64_t K) {
  if (M < 0 || K <= 0) return -1;
  const int64_t ntiles = ceil_div(K, (int64_t)TL_KB), groups = tl_grid_groups(M), nblocks = groups / TL_WAVES;
  return (nblocks + 1) * K + TL_CSC_HIST_PARTS * groups * ntiles + groups * (ntiles + 1) + 2 + 2 * (2 * groups + 1) + 16;
}

// The one-pass inspector for a CSC operand (a_indices = row indices, a_indptr = K + 1 column pointers; rows ascending inside
// every column): same outputs as spamd_spmm_tiled_inspect (blk_off[groups * (tiles + 1)], SPAMD_TILED_GROUP_ENDS; the
// stream with room for ceil(nnz / entries_per_block) + lists + slack blocks; state = one 64-bit word, non-zero on return =
// rows out of order: the lists are then left EMPTY and the caller takes the CSR route).  ws: int32 workspace of
// spamd_spmm_tiled_inspect_csc_ws(M, K) words, 8-byte aligned.
extern "C" int spamd_spmm_tiled_inspect_csc(int val_dtype, int idx_dtype, int64_t M, int64_t K, const void* a_data,
                                            const void* a_indices, const void* a_indptr, int* ws, void* state, int* blk_off,
                                            int* blocks, void* stream) {
  if (M < 0 || K <= 0) return SPAMD_EINVAL;
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  if ((uintptr_t)ws % 8) return SPAMD_EINVAL;
  const int64_t ntiles = ceil_div(K, (int64_t)TL_KB);
  if (ntiles > TL_DIRECT_MAX_TILES) return SPAMD_EINVAL;
  hipError_t e = hipMemsetAsync(state, 0, sizeof(unsigned long long), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (tl_grid_groups(M) == 0) return 0;
  SPAMD_DISPATCH_IDX(idx_dtype, I, {
    if (val_dtype == SPAMD_F32)
      return tl_launch_inspect_csc<I, float>(M, K, ntiles, (const float*)a_data, (const I*)a_indices, (const I*)a_indptr, ws,
                                             (unsigned long long*)state, blk_off, blocks, (hipStream_t)stream);
    return tl_launch_inspect_csc<I, double>(M, K, ntiles, (const double*)a_data, (const I*)a_indices, (const I*)a_indptr, ws,
                                            (unsigned long long*)state, blk_off, blocks, (hipStream_t)stream);
  })
  return SPAMD_ETYPE;
}

static int tl_set_lds_once(const void* kern) { return set_max_dynamic_lds(kern, TL_LDS); }

template <typename T, typename KERN>
static int tl_launch(KERN kern, int64_t M, int64_t K, int64_t N, const int* blocks, const int* blk_off, const T* b,
                     int64_t ldb, T* out, int64_t ldo, int touch_lines, bool group_ends, int last_cols, hipStream_t s,
                     const int* rowmap = nullptr, int64_t map_groups = 0) {
  // the 160 KB dynamic-LDS opt-in is a per-function attribute: set once per kernel, not on every multiply
  if (int rc = tl_set_lds_once(reinterpret_cast<const void*>(kern))) return rc;
  const int64_t blocks_n = (rowmap ? map_groups : tl_grid_groups(M)) / TL_WAVES;
  const int64_t npanels = N / TlFmt<T>::PANEL;
  const int64_t grid = ceil_div(blocks_n, (int64_t)8) * 8 * (npanels > 1 ? ceil_div(npanels, (int64_t)2) * 2 : 1);
  if (grid >= ((int64_t)1 << 31) || blocks_n >= ((int64_t)1 << 31)) return SPAMD_EINVAL;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TL_WAVES * 64), TL_LDS, s, M, K,
                     (int)ceil_div(K, (int64_t)TL_KB), touch_lines | (group_ends ? 1 << 16 : 0), blocks, blk_off, b, ldb, out, ldo,
                     last_cols, (int)npanels, (int)blocks_n, rowmap);
  return launch_status();
}

static int tl_product(int val_dtype, int64_t M, int64_t K, int64_t N, const int* blocks, const int* blk_off, const void* b,
                      int64_t ldb, void* out, int64_t ldo, unsigned flags, void* stream, const int* rowmap, int64_t map_groups);

extern "C" int spamd_spmm_tiled(int val_dtype, int64_t M, int64_t K, int64_t N, const int* blocks, const int* blk_off,
                                const void* b, int64_t ldb, void* out, int64_t ldo, unsigned flags, void* stream) {
  return tl_product(val_dtype, M, K, N, blocks, blk_off, b, ldb, out, ldo, flags, stream, nullptr, 0);
}

// The executor on a BALANCED layout (spamd_spmm_tiled_map_* / spamd_spmm_tiled_inspect_mapped): `groups` row groups (a
// multiple of the groups per workgroup), rowmap[groups * rows_per_group + 16] = the row of every slot (negative: unused).
// Everything else as spamd_spmm_tiled (the layout always carries per-group ends: SPAMD_TILED_GROUP_ENDS is implied).
extern "C" int spamd_spmm_tiled_mapped(int val_dtype, int64_t M, int64_t groups, int64_t K, int64_t N, const int* blocks,
                                       const int* blk_off, const int* rowmap, const void* b, int64_t ldb, void* out, int64_t ldo,
                                       unsigned flags, void* stream) {
  if (!rowmap || groups <= 0 || groups % TL_WAVES) return SPAMD_EINVAL;
  if (ldo * (val_dtype == SPAMD_F32 ? 4 : 8) >= ((int64_t)1 << 32)) return SPAMD_EINVAL;   // (32-bit row pitch in the mapped store)
  return tl_product(val_dtype, M, K, N, blocks, blk_off, b, ldb, out, ldo, flags | SPAMD_TILED_GROUP_ENDS, stream, rowmap, groups);
}

static int tl_product(int val_dtype, int64_t M, int64_t K, int64_t N, const int* blocks, const int* blk_off, const void* b,
                      int64_t ldb, void* out, int64_t ldo, unsigned flags, void* stream, const int* rowmap, int64_t map_groups) {
  if (!tl_epb(val_dtype)) return SPAMD_ETYPE;
  const int panel = val_dtype == SPAMD_F32 ? TlFmt<float>::PANEL : TlFmt<double>::PANEL;
  const int esz = val_dtype == SPAMD_F32 ? 4 : 8;
  if (M < 0 || K <= 0 || N <= 0 || N % panel != 0 || N / panel > 65535 || K / TL_KB >= ((int64_t)1 << 30)) return SPAMD_EINVAL;
  if (M == 0) return 0;
  // (rows of `out` need element alignment only: an odd float32 width puts every other row on a 4-byte boundary, which the
  // 8-byte stores of a lane's column pair take - global memory is in unaligned-access mode)
  if (((uintptr_t)b % 16) || ((ldb * esz) % 16) || ((uintptr_t)out % esz) || ((uintptr_t)blocks % 64))
    return SPAMD_EINVAL;
  if ((K + 2 * TL_KB) * ldb * esz >= ((int64_t)1 << 32)) return SPAMD_EINVAL;  // the tile DMA walks B with 32-bit byte offsets
  hipStream_t s = (hipStream_t)stream;
  const bool exact = (flags & SPAMD_EXACT_MULADD) != 0;
  const bool ends = (flags & SPAMD_TILED_GROUP_ENDS) != 0;
  const bool i32 = (flags & SPAMD_TILED_INT32) != 0;   // the stream's values, B and the result are int32 bit patterns
  if (i32 && val_dtype != SPAMD_F32) return SPAMD_EINVAL;
  // flags bits 16..23: columns of the LAST panel that are stored (0 = the whole panel).  N stays the padded width (whole
  // panels, which is what B must provide: the tile DMA reads 512 bytes of every row of B per panel); `out` then needs room for
  // N - panel + that many columns per row only (any count: an odd float32 one ends with a single-column store).
  const int last_cols = (int)((flags >> 16) & 0xffu);
  if (last_cols > panel) return SPAMD_EINVAL;
  if (ldo < N - panel + (last_cols ? last_cols : panel)) return SPAMD_EINVAL;   // a row of `out` holds every stored column
  // lines of a list that are pulled into L2 two phases ahead (flags bits 8..15; 0 = default): the launcher derives
  // it from the mean list length, longer lists pay the HBM latency on their remaining blocks
  int touch = (int)((flags >> 8) & 0xffu);
  if (touch == 0) touch = 12;
  if (touch > 64) touch = 64;
  if (val_dtype == SPAMD_F64) {
    const double* bb = (const double*)b;
    double* oo = (double*)out;
    return exact ? tl_launch<double>(&spmm_tiled_kernel<0, 5, double>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups)
                 : tl_launch<double>(&spmm_tiled_kernel<0, 4, double>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups);
  }
  const float* bb = (const float*)b;
  float* oo = (float*)out;
#ifdef SPAMD_TUNING
  // timing ablations (WRONG RESULTS by design), only in -DSPAMD_TUNING builds: 2 = no tile DMA, 5 = no fma, 6 = no LDS
  // reads/fma, 7 = neither DMA nor LDS reads nor fma.  The shipped library never reads the environment.
  const char* dbg_env = getenv("SPAMD_TILED_DBG");
  const int dbg = dbg_env ? atoi(dbg_env) : 0;
  if (dbg == 2) return tl_launch<float>(&spmm_tiled_kernel<2, 0, float>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups);
  if (dbg == 5) return tl_launch<float>(&spmm_tiled_kernel<0, 1, float>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups);
  if (dbg == 6) return tl_launch<float>(&spmm_tiled_kernel<0, 2, float>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups);
  if (dbg == 7) return tl_launch<float>(&spmm_tiled_kernel<2, 2, float>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups);
#endif
  if (i32) return tl_launch<float>(&spmm_tiled_kernel<0, 6, float>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups);
  return exact ? tl_launch<float>(&spmm_tiled_kernel<0, 3, float>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups)
               : tl_launch<float>(&spmm_tiled_kernel<0, 0, float>, M, K, N, blocks, blk_off, bb, ldb, oo, ldo, touch, ends, last_cols, s, rowmap, map_groups);
}
