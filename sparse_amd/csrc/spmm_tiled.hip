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
//     elements are re-ordered by (32-row group g, K-tile t, row, column) and written as a stream of
//     64-byte BLOCKS of eight (d0, d1) entries, d0 = (column-in-tile << 9) | (2 + 2*row-in-group),
//     d1 = value bits; every (g, t) list is padded to whole blocks with d0 = d1 = 0 (those land in a
//     junk accumulator).  blk_off[g*ntiles + t] = first block of list (g, t).
//   executor (every multiply; this kernel):
//     a workgroup of 16 waves owns 512 rows; wave w owns row group g and keeps its 32 x 128 partial
//     sums in the fixed VGPR block v[64:127] (lane l: columns 2l, 2l+1 of each row);
//     B tile t (128 x 128 fp32 = 64 KB) is copied to LDS by LDS-DMA (global_load_lds_dwordx4),
//     double-buffered, one barrier per tile;
//     the wave pulls ITS blocks with SCALAR loads (s_load_dwordx16, three blocks in a ring), so an
//     entry arrives already wave-uniform: d0 is at once the LDS row offset (v_and_or_b32 with the
//     per-lane base) and — its low 8 bits — the accumulator index for s_set_gpr_idx_idx; d1 is the
//     scalar multiplicand.  Per entry: 1 VALU address op, 1 ds_read_b64, 1 SALU, 2 v_fma_f32.
//     No readlanes, no per-row code, no divergence.
// The inner loops are generated text (tools/gen_tiled_asm.py -> spmm_tiled_asm.inc); the compiler is
// confined to v0..v23 (amdgpu_num_vgpr) and never sees v24..v127.
// Per output element the fused multiply-adds happen in the same k-ascending order as in the row-group
// kernel's FMA mode, so both kernels return bit-identical results (column indices sorted within rows).
#include "spmm_internal.h"
#include "spmm_tiled_asm.inc"
#include <stdlib.h>

namespace spamd {

constexpr int TL_RG = 32;        // rows per wave (row group)
constexpr int TL_WAVES = 16;     // waves per workgroup
#ifndef SPAMD_TL_KB
#define SPAMD_TL_KB 128
#define SPAMD_TL_NBUF 2
#endif
constexpr int TL_KB = SPAMD_TL_KB;      // B rows per tile (64 or 128: the tile must be a power of two <= 64 KB)
constexpr int TL_NBUF = SPAMD_TL_NBUF;  // LDS tile buffers: tile t+NBUF-1 is in flight while tile t is consumed
constexpr int TL_EPB = 8;        // entries per stream block
constexpr int TL_TILE = TL_KB * 512;
constexpr int TL_LDS = TL_NBUF * TL_TILE;
constexpr int TL_DMA_PER_TILE = TL_TILE / 16 / (TL_WAVES * 64);  // LDS-DMA instructions per wave per tile
static_assert(TL_LDS <= 160 * 1024 && (TL_TILE & (TL_TILE - 1)) == 0 && TL_DMA_PER_TILE >= 1, "tile geometry");
constexpr int TL_SLACK_BLOCKS = 4;  // readable blocks past the end of the stream

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
__global__ void __launch_bounds__(256) tl_blocks_kernel(const int64_t* __restrict__ seg_start, int64_t nseg,
                                                        int64_t* __restrict__ nblk) {
  GRID_STRIDE(s, nseg + 1) {
    nblk[s] = s < nseg ? (seg_start[s + 1] - seg_start[s] + TL_EPB - 1) / TL_EPB : 0;
  }
}

__global__ void __launch_bounds__(256) tl_pack_kernel(const int64_t* __restrict__ keys, const float* __restrict__ vals,
                                                      int64_t nnz, const int64_t* __restrict__ seg_start,
                                                      const int64_t* __restrict__ blk_off, int2* __restrict__ stream) {
  GRID_STRIDE(i, nnz) {
    const int64_t k = keys[i];
    const int64_t lc = k % TL_KB, r1 = k / TL_KB, lr = r1 % TL_RG, s = r1 / TL_RG;
    const int64_t dst = blk_off[s] * TL_EPB + (i - seg_start[s]);
    stream[dst] = make_int2((int)((lc << 9) | (2 + 2 * lr)), __builtin_bit_cast(int, vals[i]));
  }
}

__device__ __forceinline__ void tl_dma16(unsigned lds_base, const void* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(lds_base), "v"(src)
               : "memory", "m0");
}

template <int PK>
__device__ __forceinline__ void tl_consume(const int* blocks, int nblk, int vbase, int mask) {
  if (PK == 2)
    asm volatile(TL_ASM_CONSUME_NOFMA
                 :
                 : [ptr] "s"(blocks), [nblk] "s"(nblk), [vbase] "v"(vbase), [mask] "v"(mask)
                 : "memory", "m0", "scc", TL_CLOB_SGPR, TL_CLOB_TMP, TL_CLOB_ACC);
  else if (PK == 3)
    asm volatile(TL_ASM_CONSUME_NOLDS
                 :
                 : [ptr] "s"(blocks), [nblk] "s"(nblk), [vbase] "v"(vbase), [mask] "v"(mask)
                 : "memory", "m0", "scc", TL_CLOB_SGPR, TL_CLOB_TMP, TL_CLOB_ACC);
  else if (PK == 4)
    asm volatile(TL_ASM_CONSUME_NOSMEM
                 :
                 : [ptr] "s"(blocks), [nblk] "s"(nblk), [vbase] "v"(vbase), [mask] "v"(mask)
                 : "memory", "m0", "scc", TL_CLOB_SGPR, TL_CLOB_TMP, TL_CLOB_ACC);
  else if (PK == 5)
    asm volatile(TL_ASM_CONSUME_PIPE
                 :
                 : [ptr] "s"(blocks), [nblk] "s"(nblk), [vbase] "v"(vbase), [mask] "v"(mask)
                 : "memory", "m0", "scc", TL_CLOB_SGPR, TL_CLOB_TMP, TL_CLOB_ACC);
  else if (PK == 1)
    asm volatile(TL_ASM_CONSUME_PK
                 :
                 : [ptr] "s"(blocks), [nblk] "s"(nblk), [vbase] "v"(vbase), [mask] "v"(mask)
                 : "memory", "m0", "scc", TL_CLOB_SGPR, TL_CLOB_TMP, TL_CLOB_ACC);
  else
    asm volatile(TL_ASM_CONSUME
                 :
                 : [ptr] "s"(blocks), [nblk] "s"(nblk), [vbase] "v"(vbase), [mask] "v"(mask)
                 : "memory", "m0", "scc", TL_CLOB_SGPR, TL_CLOB_TMP, TL_CLOB_ACC);
}

// DBG (timing ablations only): 1 = no consume, 2 = no tile DMA.  PK: v_pk_fma_f32 instead of 2 v_fma_f32.
template <int DBG, int PK>
__global__ void __launch_bounds__(TL_WAVES * 64) __attribute__((amdgpu_num_vgpr(24)))
spmm_tiled_kernel(int64_t M, int64_t K, int64_t ntiles, const int* __restrict__ stream,
                  const int* __restrict__ blk_off, const float* __restrict__ b, int64_t ldb,
                  float* __restrict__ out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // the only LDS object: starts at LDS byte 0
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = uniform(tid >> 6);
  const int64_t g = (int64_t)blockIdx.x * TL_WAVES + wv;  // my row group (lists exist for every wave of the grid)
  b += (int64_t)blockIdx.y * 128;                          // column panel of B and of the result
  out += (int64_t)blockIdx.y * 128;

  asm volatile(TL_ASM_ZERO ::: "memory", TL_CLOB_ACC);

  // Tile DMA: every wave issues exactly TL_DMA_PER_TILE instructions per tile, all lanes active — so
  // vmcnt arithmetic is exact.  Full tiles: per-thread source pointers advanced by one tile per call
  // (no 64-bit multiplies in the loop).  The last, partial tile (and the dummy re-issues past the end
  // that keep the per-iteration count fixed): rows past K are clamped to row K-1, which no entry refers to.
  const int64_t nfull = K / TL_KB;
  const int64_t tile_step_bytes = (int64_t)TL_KB * ldb * 4;
  const char* src[TL_DMA_PER_TILE];
#pragma unroll
  for (int i = 0; i < TL_DMA_PER_TILE; ++i) {
    const int e = (i * (TL_WAVES * 64) + tid) * 4;
    src[i] = reinterpret_cast<const char*>(b + (int64_t)(e >> 7) * ldb + (e & 127));
  }
  int64_t next_tile = 0;  // issue_tile is called for t = 0, 1, 2, ... in order
  auto issue_tile = [&]() {
    const int64_t t = next_tile++;
    const unsigned buf = (unsigned)(t % TL_NBUF) * TL_TILE;
    if (t < nfull) {
#pragma unroll
      for (int i = 0; i < TL_DMA_PER_TILE; ++i) {
        tl_dma16(buf + (unsigned)(i * TL_WAVES + wv) * 1024u, src[i]);
        src[i] += tile_step_bytes;
      }
    } else {
      const int64_t kb0 = (t < ntiles ? t : ntiles - 1) * TL_KB;
#pragma unroll
      for (int i = 0; i < TL_DMA_PER_TILE; ++i) {
        const int e = (i * (TL_WAVES * 64) + tid) * 4;
        int64_t r = kb0 + (e >> 7);
        if (r >= K) r = K - 1;
        tl_dma16(buf + (unsigned)(i * TL_WAVES + wv) * 1024u, b + r * ldb + (e & 127));
      }
    }
  };

  if (DBG != 2)
    for (int t = 0; t < TL_NBUF - 1; ++t) issue_tile();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TL_NBUF - 2) * TL_DMA_PER_TILE) : "memory");
  __syncthreads();

  // The block stream is read once, by scalar loads: no hardware prefetcher, and every s_waitcnt on the
  // LDS reads (lgkmcnt(0): SMEM returns out of order) also waits for the scalar load issued a round
  // earlier, so its latency must be an L2 hit (~270 cycles), never HBM (~1-2 us: measured 2.25 ms
  // without this).  Two tile phases ahead, the consuming wave touches the 64-byte lines of that list
  // with one vector load (lane i -> line i, result discarded in v61): HBM -> this XCD's L2.
  // (A further scalar-cache prefetch stage was measured slower: K$ hits still cost ~160 cycles and
  // the scalar return path is 4 B/clk per CU: tools/micro/smem_lat.hip.)
  auto touch_lines = [&](const void* p, int64_t nlines) {  // always ONE instruction (vmcnt arithmetic)
    const int64_t l = lane < nlines ? lane : 0;
    asm volatile("global_load_dword v61, %0, off" ::"v"(reinterpret_cast<const char*>(p) + l * 64) : "memory", "v61");
  };
  // List boundaries of this wave: blk_off[g*ntiles + t], t = 0..ntiles.  64 of them at a time live in one
  // VGPR (lane <-> tile) and are read with v_readlane: no memory access per tile (a per-tile load would
  // make the compiler wait vmcnt(0), i.e. for the tile DMA in flight).
  const int* const myoff = blk_off + g * ntiles;
  int64_t obase = 0;
  auto load_offsets = [&](int64_t base) {
    const int64_t q = base + lane;
    return myoff[q < ntiles ? q : ntiles];
  };
  int offreg = load_offsets(0);
  asm volatile("" : "+v"(offreg));
  auto list_start = [&](int64_t t) -> int64_t {  // needs obase <= t < obase + 64 (or t > ntiles: clamped)
    const int64_t q = t < ntiles ? t : ntiles;
    return (int64_t)wave_bcast(offreg, (int)(q - obase));
  };
  int64_t o0 = list_start(0), o1 = list_start(1), o2 = list_start(2);  // starts of lists t, t+1, t+2
  touch_lines(stream + o0 * (TL_EPB * 2), o2 - o0);
  for (int64_t t = 0; t < ntiles; ++t) {
    if (DBG != 2) issue_tile();  // tile t + NBUF - 1
    const int nblk = (int)(o1 - o0);
    if (nblk > 0 && DBG != 1)
      tl_consume<PK>(stream + o0 * (TL_EPB * 2), nblk, (int)((unsigned)(t % TL_NBUF) * TL_TILE) + lane * 8,
                     (int)0xfffffe00);
    if (t + 3 <= ntiles && t + 3 - obase >= 64) {  // once per 61 tiles (this load does wait for the DMA)
      obase = t + 3;
      offreg = load_offsets(obase);
      asm volatile("" : "+v"(offreg));  // the wait for this load stays inside the branch
    }
    const int64_t o3 = list_start(t + 3);
    touch_lines(stream + o2 * (TL_EPB * 2), DBG != 4 ? o3 - o2 : 0);
    o0 = o1;
    o1 = o2;
    o2 = o3;
    // tile t+1 (this wave's share) has landed: everything but the newest NBUF-2 tiles and one touch
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TL_NBUF - 2) * (TL_DMA_PER_TILE + 1) + 1) : "memory");
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // write my rows
  const int64_t row0 = g * TL_RG;
  const int64_t left = M - row0;
  const int nvalid = left <= 0 ? 0 : (left < TL_RG ? (int)left : TL_RG);
  float* const obase_p = out + row0 * ldo + lane * 2;
  const int64_t stride_bytes = ldo * 4;
  asm volatile(TL_ASM_STORE
               :
               : [lo] "v"((unsigned)((uintptr_t)obase_p & 0xffffffffu)), [hi] "v"((unsigned)((uintptr_t)obase_p >> 32)),
                 [stride] "s"(stride_bytes), [n] "s"(nvalid)
               : "memory", "scc", "s36", "v60", "v61", TL_CLOB_ACC);
}

static int64_t tl_grid_groups(int64_t M) { return ceil_div(ceil_div(M, (int64_t)TL_RG), (int64_t)TL_WAVES) * TL_WAVES; }

}  // namespace spamd

using namespace spamd;

extern "C" int spamd_spmm_tiled_params(int* rows_per_group, int* tile_rows, int* groups_per_block,
                                       int* entries_per_block, int* slack_blocks) {
  if (rows_per_group) *rows_per_group = TL_RG;
  if (tile_rows) *tile_rows = TL_KB;
  if (groups_per_block) *groups_per_block = TL_WAVES;
  if (entries_per_block) *entries_per_block = TL_EPB;
  if (slack_blocks) *slack_blocks = TL_SLACK_BLOCKS;
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

extern "C" int spamd_spmm_tiled_lists(int64_t nnz, const int64_t* tiled_keys_sorted, int64_t M, int64_t K,
                                      int64_t* seg_start, int64_t* nblk, void* stream) {
  if (nnz < 0 || M < 0 || K <= 0) return SPAMD_EINVAL;
  const int64_t nseg = tl_grid_groups(M) * ceil_div(K, (int64_t)TL_KB);
  hipLaunchKernelGGL(tl_seg_start_kernel, dim3(tl_blocks_for(nnz + 1)), dim3(256), 0, (hipStream_t)stream,
                     tiled_keys_sorted, nnz, nseg, seg_start);
  hipLaunchKernelGGL(tl_blocks_kernel, dim3(tl_blocks_for(nseg + 1)), dim3(256), 0, (hipStream_t)stream, seg_start,
                     nseg, nblk);
  return launch_status();
}

extern "C" int spamd_spmm_tiled_pack(int64_t nnz, const int64_t* tiled_keys_sorted, const float* vals_sorted,
                                     const int64_t* seg_start, const int64_t* blk_off, int64_t total_blocks,
                                     int* blocks, void* stream) {
  if (nnz < 0 || total_blocks < 0) return SPAMD_EINVAL;
  hipError_t e = hipMemsetAsync(blocks, 0, (size_t)(total_blocks + TL_SLACK_BLOCKS) * TL_EPB * 8, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (nnz == 0) return 0;
  hipLaunchKernelGGL(tl_pack_kernel, dim3(tl_blocks_for(nnz)), dim3(256), 0, (hipStream_t)stream, tiled_keys_sorted,
                     vals_sorted, nnz, seg_start, blk_off, reinterpret_cast<int2*>(blocks));
  return launch_status();
}

extern "C" int spamd_spmm_tiled(int64_t M, int64_t K, int64_t N, const int* blocks, const int* blk_off,
                                const float* b, int64_t ldb, float* out, int64_t ldo, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || N % 128 != 0 || N / 128 > 65535) return SPAMD_EINVAL;
  if (M == 0) return 0;
  if (((uintptr_t)b % 16) || (ldb % 4) || ((uintptr_t)out % 8) || (ldo % 2) || ((uintptr_t)blocks % 64))
    return SPAMD_EINVAL;
  const char* dbg_env = getenv("SPAMD_TILED_DBG");  // timing ablations: 1 = no consume, 2 = no tile DMA, 4 = no stream touch, 8 = two v_fma_f32 instead of v_pk_fma_f32
  const int dbg = dbg_env ? atoi(dbg_env) : 0;
  auto kern = dbg == 1 ? &spmm_tiled_kernel<1, 0>
            : dbg == 2 ? &spmm_tiled_kernel<2, 0>
            : dbg == 4 ? &spmm_tiled_kernel<4, 0>
            : dbg == 8 ? &spmm_tiled_kernel<0, 0>
            : dbg == 9 ? &spmm_tiled_kernel<0, 4>
            : dbg == 10 ? &spmm_tiled_kernel<2, 4>
            : dbg == 5 ? &spmm_tiled_kernel<0, 2>
            : dbg == 6 ? &spmm_tiled_kernel<0, 3>
            : dbg == 7 ? &spmm_tiled_kernel<2, 3>
            : dbg == 11 ? &spmm_tiled_kernel<0, 1>
            : dbg == 12 ? &spmm_tiled_kernel<2, 5> : &spmm_tiled_kernel<0, 5>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     TL_LDS);
  if (e != hipSuccess) return (int)e;
  const int64_t blocks_n = tl_grid_groups(M) / TL_WAVES;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks_n, (unsigned)(N / 128)), dim3(TL_WAVES * 64), TL_LDS, (hipStream_t)stream, M, K,
                     ceil_div(K, (int64_t)TL_KB), blocks, blk_off, b, ldb, out, ldo);
  return launch_status();
}
