// gather -> small GEMM -> scatter over a device rulebook (gfx950, fp32 MFMA).
//
// One output-stationary kernel serves submanifold conv, strided conv,
// deconvolution and (with transposed weights) every dInput:
//   * a workgroup owns M consecutive OUTPUT rows x COB output planes, kept in LDS;
//   * because every rulebook list is sorted on its output column, the pairs of
//     list k that land in the tile are one contiguous segment [bases[k][g0], bases[k][g1]);
//   * per (k, Ci-chunk) the segment's input rows are gathered with 16-byte
//     coalesced loads into an LDS tile, each wave multiplies 16-pair groups by its
//     16-plane slice of W[k] with v_mfma_f32_16x16x4_f32 (weights live in registers
//     as the B operand) and adds the 16x16 result into the LDS output tile;
//   * each output element is owned by one wave and visited in (k, pair) order:
//     no global atomics, results are bitwise reproducible; every output row is
//     written to HBM exactly once.
// The reference instead stages 16x16 tiles for CUDA-core FMAs and scatters with
// global atomicAdd (sparseconvnet/SCN/CUDA/Convolution.cu:1059-1152).
//
// dW[k] = In^T . dOut over the pairs of list k: split-K over pair chunks, both
// row sets staged in LDS, 16x16x4 MFMA with the pair index as the K dimension,
// float atomics into dW (as the reference, sparseconvnet/SCN/CUDA/Convolution.cu:749).
#include "scn_common.h"
#include "tile4_loop_asm.inc"  // k_conv_tile4's pass loop in assembly (generated: tools/gen_tile4_loop.py)
#ifdef SCN_TUNING
#include "tile4_loop_asm_tuning.inc"  // ... and its timing-only ablations
#endif

#include <type_traits>

namespace scn {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Where a gather-mm launch leaves per-block column sums for the batch norm next to it (scn_conv_forward's `colstats`):
//   x == nullptr: sum v and sum v^2 of the RESULT v (the batch norm that CONSUMES this tensor: forward statistics);
//   x != nullptr: the launch computes the gradient d w.r.t. the OUTPUT y = leaky(fma(x, w, b)) of a batch norm over x
//                 (it is the dInput of the convolution that consumed y): sum d' and sum (x - mean) d' with
//                 d' = d * (y > 0 ? 1 : leak) -- the two reductions of the batch norm's backward pass
//                 (sparseconvnet/SCN/CPU/BatchNormalization.cpp:75-87), so that pass need not read x and d once more.
struct StatsArg {
  float *out;
  const float *x, *mean, *invstd, *gamma, *beta;
  float leak;
  // (rides along to every gather-mm kernel:) leading dimensions, in floats, of the launch's `in` and `out` when they are
  // column slices of wider row-major tensors (0 = contiguous: C_i / C_o).  ldi: k_conv_tile4, k_conv_tile2, k_conv_expand;
  // ldo: k_conv_expand only (scn_conv_forward checks)
  int ldi, ldo;
};
// per-lane constants of four consecutive planes c .. c+3 and the two sums for one float4 of the result
struct StatsCols {
  floatx4 m, w, b;
  float leak;
  bool bn;
  __device__ __forceinline__ void init(const StatsArg &a, int c) {
    bn = a.x != nullptr;
    leak = a.leak;
    if (bn) {
      m = *(const floatx4 *)(a.mean + c);
      const floatx4 is = *(const floatx4 *)(a.invstd + c);
      const floatx4 g = a.gamma ? *(const floatx4 *)(a.gamma + c) : floatx4{1.f, 1.f, 1.f, 1.f};
      const floatx4 be = a.beta ? *(const floatx4 *)(a.beta + c) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // the forward pass's own arithmetic (k_bn_finalize): bit-identical sign of y
        w[e] = is[e] * g[e];
        b[e] = fmaf(-m[e], w[e], be[e]);
      }
    }
  }
  // v = four results of row `row_x` (pointer to the row's planes c .. c+3 of x when bn)
  __device__ __forceinline__ void add(const floatx4 &v, const float *xrow, floatx4 &s0, floatx4 &s1) const {
    if (!bn) {
      s0 += v;
      s1 += v * v;
    } else {
      const floatx4 xv = *(const floatx4 *)xrow;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = (leak == 1.f || fmaf(xv[e], w[e], b[e]) > 0.f) ? v[e] : v[e] * leak;
        s0[e] += d;
        s1[e] += (xv[e] - m[e]) * d;
      }
    }
  }
};

constexpr int KMAX = 27;
constexpr int SCN_TILE4_MAX_CHUNKS = 6;  // k_conv_tile4: C_i <= 384 (pass table in LDS: 35.1 + 5.2 KB = four workgroups per CU, to the byte)

// diagnostic build only (DBG bit 8 = 256): per-segment cycle sums of the pipelined kernel, added from
// lane 0 of every wave at kernel end into a buffer nothing else reads (scn_debug_stamps copies it out).
// Workgroup barrier that orders LDS traffic only: __syncthreads() is fence + s_barrier and hipcc turns the
// fence into s_waitcnt vmcnt(0) whenever global loads are in flight -- here that would drain the row
// gathers that are deliberately left outstanding across the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ unsigned long long g_stamps[8];
#ifdef SCN_TUNING
__device__ int g_sk_prio_dev = 1;  // tuning key 20: 0 = stream-K launches without the alternating wave priority (A/B)
constexpr bool SK_TUNING_PRIO = true;
#else
constexpr int g_sk_prio_dev = 1;
constexpr bool SK_TUNING_PRIO = false;
#endif
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}


// ---------------------------------------------------------------- forward ---
template <int NW, int COS, int M, int CK>
__global__ void __launch_bounds__(64 * NW)
    k_conv_tile(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ W,
                const float *__restrict__ bias, const float *__restrict__ residual, StatsArg colstats, const int2 *__restrict__ rules,
                const int32_t *__restrict__ bases, int K, int64_t stride, int in_col,
                int64_t n_out, int n_gran, int Ci, int Co) {
  constexpr int COB = NW * COS * 16;
  constexpr int NT = 64 * NW;
  constexpr int AS = CK + 4;   // a_tile row stride (floats): conflict-free ds_read_b128 across 16 rows
  constexpr int OS = COB + 4;  // o_tile row stride
  __shared__ __attribute__((aligned(16))) float a_tile[M * AS];
  __shared__ __attribute__((aligned(16))) float o_tile[M * OS];
  __shared__ int ol[M];
  __shared__ int seg[KMAX + 1][2];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int64_t o0 = (int64_t)blockIdx.x * M;
  const int co0 = blockIdx.y * COB;
  const int g0 = blockIdx.x * (M / SCN_TILE_GRANULE);
  const int g1 = min(g0 + M / SCN_TILE_GRANULE, n_gran);

  for (int i = tid; i < M * COB; i += NT) {
    const int r = i / COB, c = i % COB;
    float v = bias ? bias[co0 + c] : 0.f;
    if (residual && o0 + r < n_out) v += residual[(o0 + r) * Co + co0 + c];
    o_tile[r * OS + c] = v;
  }
  if (tid < K) {
    seg[tid][0] = bases[(int64_t)tid * (n_gran + 1) + g0];
    seg[tid][1] = bases[(int64_t)tid * (n_gran + 1) + g1];
  }
  __syncthreads();

  const int li = lane & 15;  // A row / B column / D column inside a 16x16 tile
  const int lq = lane >> 4;  // k-quad of the A/B operands, row-quad of D

  for (int k = 0; k < K; ++k) {
    const int s = seg[k][0];
    const int c = seg[k][1] - s;
    if (c <= 0) continue;
    const int2 *rk = rules + (int64_t)k * stride + s;
    const int ngroups = (c + 15) >> 4;
    for (int c0 = 0; c0 < Ci; c0 += CK) {
      const int ck = min(CK, Ci - c0);  // multiple of 16
      const int ch_per_row = ck >> 2;
      __syncthreads();  // previous phase finished reading a_tile / ol
      for (int idx = tid; idx < c * ch_per_row; idx += NT) {
        const int row = idx / ch_per_row, ch = idx - row * ch_per_row;
        const int2 p = rk[row];
        const int src = in_col ? p.y : p.x;
        const float4 v = *(const float4 *)(in + (int64_t)src * Ci + c0 + ch * 4);
        *(float4 *)(a_tile + row * AS + ch * 4) = v;
        if (c0 == 0 && ch == 0) ol[row] = (int)((in_col ? p.x : p.y) - o0);
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < COS; ++q) {
        const int cw = (w * COS + q) * 16;  // plane offset inside the block's COB
        // B operand: lane holds W[k][c0 + 16 s' + 4 lq + e][co0 + cw + li]
        float breg[CK / 4];
        const float *wp = W + ((int64_t)k * Ci + c0 + 4 * lq) * Co + co0 + cw + li;
#pragma unroll
        for (int sp = 0; sp < CK / 16; ++sp)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            breg[sp * 4 + e] = (sp * 16 < ck) ? wp[(int64_t)(sp * 16 + e) * Co] : 0.f;
        for (int g = 0; g < ngroups; ++g) {
          floatx4 acc = {0.f, 0.f, 0.f, 0.f};
          const float *ap = a_tile + (g * 16 + li) * AS + 4 * lq;
#pragma unroll
          for (int sp = 0; sp < CK / 16; ++sp) {
            if (sp * 16 < ck) {
              const float4 a4 = *(const float4 *)(ap + sp * 16);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, breg[sp * 4 + 0], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, breg[sp * 4 + 1], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, breg[sp * 4 + 2], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, breg[sp * 4 + 3], acc, 0, 0, 0);
            }
          }
          // D: lane holds rows (pairs) g*16 + 4 lq + r, column li
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = g * 16 + 4 * lq + r;
            if (row < c) o_tile[ol[row] * OS + cw + li] += acc[r];
          }
        }
      }
    }
  }
  __syncthreads();
  const int rows = (int)min((int64_t)M, n_out - o0);
  for (int i = tid; i < rows * (COB / 4); i += NT) {
    const int r = i / (COB / 4), c4 = i % (COB / 4);
    *(float4 *)(out + (o0 + r) * Co + co0 + c4 * 4) = *(const float4 *)(o_tile + r * OS + c4 * 4);
  }
}

// ------------------------------------------------------ weight packing ---
// Source element (k, r, c) = reduction plane r, output plane c of list k:
//   transposed ? W[ks][c][r] : W[ks][r][c],  ks = reverse ? K-1-k : k   (W row-major [K][rows][cols]).
// mode 0: plain   Wp[k][r][c]
// mode 1: MFMA B-fragment order Wp[k][r/16][c/16][lq][li][e] = (k, 16(r/16) + 4 lq + e, 16(c/16) + li):
//         a wave reads the 16x16 block it needs for 4 consecutive mfma_16x16x4 as ONE coalesced 1 KiB load.
__global__ void __launch_bounds__(256)
    k_weight_pack(const float *__restrict__ W, float *__restrict__ Wp, int K, int Cr, int Cc,
                  int transposed, int reverse, int mode) {
  const int64_t total = (int64_t)K * Cr * Cc;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int k, r, c;
    if (mode == 0) {
      k = (int)(t / ((int64_t)Cr * Cc));
      const int rem = (int)(t % ((int64_t)Cr * Cc));
      r = rem / Cc; c = rem % Cc;
    } else {
      int64_t u = t;
      const int e = (int)(u & 3); u >>= 2;
      const int li = (int)(u & 15); u >>= 4;
      const int lq = (int)(u & 3); u >>= 2;
      const int cs = (int)(u % (Cc / 16)); u /= (Cc / 16);
      const int rs = (int)(u % (Cr / 16)); u /= (Cr / 16);
      k = (int)u;
      r = rs * 16 + lq * 4 + e;
      c = cs * 16 + li;
    }
    const int ks = reverse ? K - 1 - k : k;
    Wp[t] = transposed ? W[((int64_t)ks * Cc + c) * Cr + r] : W[((int64_t)ks * Cr + r) * Cc + c];
  }
}

// Every (weight, orientation) pair of a network re-laid out by ONE launch (scn_weight_pack_batch): the reference reads the
// module's weights on every call (sparseconvnet/SCN/CUDA/Convolution.cpp:105-162); here they are re-read once per
// optimizer step / forward pass, all of them together.  Descriptor i covers blocks [first_block_i, first_block_{i+1}).
struct WPackDesc {
  const float *W;
  float *Wp;
  int32_t K, Cr, Cc, flags;  // flags: bit 0 transposed, bit 1 reverse, bit 2 plain (mode 0)
  int64_t first_block;
};
constexpr int WPB_ELEMS = 2048;  // elements per block of k_weight_pack_multi
__global__ void __launch_bounds__(256)
    k_weight_pack_multi(const WPackDesc *__restrict__ descs, int n) {
  // uniform binary search: the last descriptor whose first block is <= this block
  int lo = 0, hi = n - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const WPackDesc d = descs[lo];
  const int K = d.K, Cr = d.Cr, Cc = d.Cc;
  const bool transposed = d.flags & 1, reverse = d.flags & 2, plain = d.flags & 4;
  const int64_t total = (int64_t)K * Cr * Cc;
  const int64_t t0 = (b - d.first_block) * WPB_ELEMS + threadIdx.x;
  const int ncs = Cc / 16, nrs = Cr / 16;
#pragma unroll
  for (int j = 0; j < WPB_ELEMS / 256; ++j) {
    const int64_t t = t0 + j * 256;
    if (t >= total) break;
    int k, r, c;
    if (plain) {
      k = (int)(t / ((int64_t)Cr * Cc));
      const int rem = (int)(t % ((int64_t)Cr * Cc));
      r = rem / Cc; c = rem % Cc;
    } else {
      unsigned u = (unsigned)(t >> 8);  // 256 elements per 16 x 16 block
      const int e = (int)(t & 3), li = (int)((t >> 2) & 15), lq = (int)((t >> 6) & 3);
      const int cs = (int)(u % (unsigned)ncs); u /= (unsigned)ncs;
      const int rs = (int)(u % (unsigned)nrs); u /= (unsigned)nrs;
      k = (int)u;
      r = rs * 16 + lq * 4 + e;
      c = cs * 16 + li;
    }
    const int ks = reverse ? K - 1 - k : k;
    d.Wp[t] = transposed ? d.W[((int64_t)ks * Cc + c) * Cr + r] : d.W[((int64_t)ks * Cr + r) * Cc + c];
  }
}

// ------------------------------------------------- forward, pipelined (v2) ---
// Same decomposition as k_conv_tile, restructured so that the matrix pipe does not wait for HBM:
//   * M = 32 NW output rows per workgroup (128 for 4 waves): fewer padded pair groups, W[k] fetched once
//     per 128 rows;
//   * the gathered rows of phase p+1 and the weight fragments of phase p+1 are requested into registers
//     BEFORE the MFMAs of phase p start and are written to LDS after them (issue-early / write-late);
//     the pair indices are fetched two lists ahead (triple-buffered in LDS), so no phase waits on a
//     dependent index -> row load chain;
//   * weights come pre-packed in fragment order (k_weight_pack mode 1): one 16-byte load per lane per
//     16 input planes;
//   * workgroups are renumbered so that each XCD owns a contiguous range of tiles: neighbouring tiles
//     gather overlapping rows and now share one L2.
template <int NW, int COS, int CK, int DBG = 0, int NWM = 1, int M_ = 0>
__global__ void __launch_bounds__(64 * NW * NWM) __attribute__((amdgpu_waves_per_eu(2, 8)))  // (<= 256 registers: accumulators stay in VGPRs, no v_accvgpr_read before the output-tile adds)
    k_conv_tile2(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ Wp,
                 const float *__restrict__ bias, const float *__restrict__ residual, StatsArg colstats, const int2 *__restrict__ rules,
                 const int32_t *__restrict__ bases, int K, int64_t stride, int in_col,
                 int64_t n_out, int n_gran, int Ci, int Co, int ntiles, int ncb, int ksplit,
                 unsigned in_bytes, unsigned wp_bytes, float *__restrict__ final_out = nullptr, unsigned *tickets = nullptr) {
  // requires Ci % CK == 0 (the launcher picks CK as the largest of 64/32/16 dividing Ci)
  constexpr int M = M_ ? M_ : 32 * NW;
  constexpr int COB = NW * COS * 16;
  constexpr int NT = 64 * NW * NWM;  // NWM waves share each plane slice and split the pair groups
  // a_tile: plain rows padded by 4 floats (row stride 68 floats = 4 banks mod 64: the 16 rows a
  // ds_read_b128 lane group touches cover all 64 banks).  A variant split into 4 planes, one per MFMA
  // k-quad, makes the reads fully conflict free but was slower overall (QP below).
  constexpr bool QP = (DBG & 128) != 0;  // opt-in for A/B runs only: measured 10 % SLOWER (the 16-byte staging
                                         // writes of one row then hit 4 planes on the same banks)
  constexpr int AQ = QP ? CK / 4 : CK;  // floats per row inside one plane (16 for CK 64)
  constexpr int AS = AQ + 4;            // padded row stride inside a plane
  constexpr int APL = QP ? M * AS : 0;  // plane stride
  constexpr int OS = COB + 4;
  constexpr int CH = CK / 4;            // 16-byte chunks per staged row
  constexpr int NSLOT = M * CH / NT;    // row-chunks each thread stages per phase (= CK/8)
  constexpr int NIDX = (M + NT - 1) / NT;
  constexpr int RPS = NT / CH;          // rows between two slots of one thread
  constexpr int NS16 = CK / 16;
  __shared__ __attribute__((aligned(16))) float a_tile[QP ? 4 * APL : M * AS];
  __shared__ __attribute__((aligned(16))) float o_tile[(M + 1) * OS];  // row M: sink for padded pairs
  __shared__ unsigned srcbuf[4][M];  // byte offset of each pair's input row; slot-major so a thread's NSLOT
                                     // offsets are contiguous: index(row) = (row % RPS) * NSLOT + row / RPS
  __shared__ int olbuf[4][M];
  __shared__ int klist[KMAX];
  __shared__ int kseg[KMAX][2];
  __shared__ int nk_s;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave index: uniform, but derived from threadIdx -> divergent to the compiler unless made scalar
  const int w = __builtin_amdgcn_readfirstlane((tid >> 6) % NW);   // plane slice of this wave
  const int wm = __builtin_amdgcn_readfirstlane((tid >> 6) / NW);  // which share of the pair groups
  const int li = lane & 15, lq = lane >> 4;
  // Gathered rows and weight fragments are fetched with buffer loads: the per-lane part of the address
  // is ONE 32-bit add (row byte offsets are precomputed once per pair into LDS), everything else is scalar.
  // With flat 64-bit addressing the address arithmetic of the 12 loads a wave issues per phase took
  // 27-34 % of its cycles (in-kernel stamps, tools/stamps.py).
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Wp, 0, (int)wp_bytes, 0x00020000);

  // XCD-aware renumbering (bijective): blocks b, b+8, .. share an XCD; give each a contiguous tile range
  int lin;
  {
    const int total = ntiles * ncb;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int q = total >> 3, r = total & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile = lin / ncb;
  const int co0 = (lin % ncb) * COB;
  const int64_t o0 = (int64_t)tile * M;
  const int g0 = tile * (M / SCN_TILE_GRANULE);
  const int g1 = min(g0 + M / SCN_TILE_GRANULE, n_gran);

  // residual: the tile starts from the rows of another tensor (out = residual + bias + sum ...): the shortcut of a
  // residual block is added here for the price of reading it, instead of by a separate elementwise kernel.  Eight
  // elements per thread are requested together (one HBM round trip per eight instead of one per element).
  for (int i0 = tid; i0 < M * COB; i0 += 8 * NT) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * NT, r = i / COB, c = i % COB;
      v[u] = (residual && blockIdx.y == 0 && i < M * COB && o0 + r < n_out) ? residual[(o0 + r) * Co + co0 + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * NT, r = i / COB, c = i % COB;
      if (i < M * COB) o_tile[r * OS + c] = v[u] + ((bias && blockIdx.y == 0) ? bias[co0 + c] : 0.f);
    }
  }
  if (tid < 64) {
    // one lane per kernel offset: segment of list k inside this tile, then a ballot-compacted list of
    // the non-empty offsets (K <= 27 < 64)
    int s = 0, e = 0;
    if (tid < K) {
      s = bases[(int64_t)tid * (n_gran + 1) + g0];
      e = bases[(int64_t)tid * (n_gran + 1) + g1];
    }
    // ksplit > 1 (coarse levels): this workgroup takes the offsets k == blockIdx.y (mod ksplit)
    const bool mine = e > s && (tid % ksplit) == (int)blockIdx.y;
    const unsigned long long m = __ballot(mine);
    if (mine) {
      const int pos = __popcll(m & ((1ull << tid) - 1ull));
      klist[pos] = tid;
      kseg[pos][0] = s;
      kseg[pos][1] = e - s;
    }
    if (tid == 0) nk_s = __popcll(m);
  }
  __syncthreads();
  // values read back from LDS are wave-uniform but hipcc cannot know: without readfirstlane every buffer
  // load whose scalar offset derives from them is wrapped in a waterfall loop (readfirstlane + saveexec)
  auto uni = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  const int nk = uni(nk_s);
  const int nch = Ci / CK;
  const int nphase = nk * nch;

  // Branch-free staging: rows past the end of a segment re-read its last pair (same cache lines) and
  // scatter into the sink row M, so no lane ever diverges inside a phase.
  auto load_indices = [&](int kk, int2 (&pr)[NIDX]) __attribute__((always_inline)) {
    const int2 *rk = rules + (int64_t)uni(klist[kk]) * stride + uni(kseg[kk][0]);
    const int c = uni(kseg[kk][1]);
#pragma unroll
    for (int j = 0; j < NIDX; ++j) pr[j] = rk[min(tid + j * NT, c - 1)];
  };
  auto store_indices = [&](int kk, const int2 (&pr)[NIDX]) __attribute__((always_inline)) {
    const int c = uni(kseg[kk][1]);
    const int b = kk & 3;
#pragma unroll
    for (int j = 0; j < NIDX; ++j) {
      const int r = tid + j * NT;
      if (r < M) {
        srcbuf[b][(r % RPS) * NSLOT + r / RPS] = (unsigned)(in_col ? pr[j].y : pr[j].x) * (unsigned)((colstats.ldi ? colstats.ldi : Ci) * 4);
        olbuf[b][r] = (r < c) ? (int)((in_col ? pr[j].x : pr[j].y) - o0) : M;
      }
    }
  };
  floatx4 preA[NSLOT];  // gathered rows in flight, one phase ahead
  // slot j of a thread covers row (tid + j NT) / CH; with CH * 16 == NT (CK 64, 4 waves) that is pair
  // group j, so "j < groups of this list" is wave-uniform and unused groups cost nothing.
  constexpr bool SLOT_IS_GROUP = (NT == CH * 16) && (DBG & 4);
  const unsigned lane_chunk = (unsigned)(tid % CH) * 16u;  // this thread's 16-byte chunk inside a staged row
  auto read_offsets = [&](int p, unsigned (&off)[NSLOT]) __attribute__((always_inline)) {
    const unsigned *sb = srcbuf[(p / nch) & 3] + (tid / CH) * NSLOT;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) off[j] = sb[j];
  };
  auto issue_rows_at = [&](int p, const unsigned (&off)[NSLOT], floatx4 (&pre)[NSLOT]) __attribute__((always_inline)) {
    const int kk = p / nch, c0 = (p - kk * nch) * CK;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      if (DBG & 2) {
        pre[j] = floatx4{1.f, 2.f, 3.f, (float)off[j]};
      } else {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off[j] + lane_chunk, c0 * 4, 0);
        pre[j] = __builtin_bit_cast(floatx4, v);
      }
    }
  };
  auto issue_rows = [&](int p, floatx4 (&pre)[NSLOT]) __attribute__((always_inline)) {
    unsigned off[NSLOT];
    read_offsets(p, off);
    issue_rows_at(p, off, pre);
  };
  auto store_rows = [&](int p, const floatx4 (&pre)[NSLOT]) __attribute__((always_inline)) {
    const int ng = (uni(kseg[p / nch][1]) + 15) >> 4;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int idx = tid + j * NT;
      if (DBG & 16) asm volatile("" ::"v"(pre[j]));
      else if (!SLOT_IS_GROUP || j < ng)
        *(floatx4 *)(a_tile + (QP ? ((idx % CH) & 3) * APL + (idx / CH) * AS + ((idx % CH) >> 2) * 4
                                  : (idx / CH) * AS + (idx % CH) * 4)) = pre[j];
    }
  };
  floatx4 wcur[COS][NS16], wnext[COS][NS16];
  auto issue_weights = [&](int p, floatx4 (&wr)[COS][NS16]) __attribute__((always_inline)) {
    const int kk = p / nch, c0 = (p - kk * nch) * CK;
    const int k = uni(klist[kk]);
    // Wp[k][r/16][c/16][lane][4] floats: scalar block offset + lane * 16 bytes
    const int blk0 = (k * (Ci / 16) + c0 / 16) * (Co / 16) + co0 / 16 + w * COS;
#pragma unroll
    for (int q = 0; q < COS; ++q)
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wp_rsrc, (unsigned)lane * 16u,
                                                              (blk0 + sp * (Co / 16) + q) * 1024, 0);
        wr[q][sp] = __builtin_bit_cast(floatx4, v);
      }
  };
  // One 16-pair group x COS plane slices.  Operands are SWAPPED (A = weight fragment, B = gathered rows):
  // D[i = plane][j = pair], so a lane ends up with 4 consecutive output planes of ONE pair and the
  // accumulate into the LDS output tile is a single 16-byte read-add-write per slice.
  auto load_a = [&](int g, floatx4 (&a)[NS16]) __attribute__((always_inline)) {
    const float *ap = QP ? a_tile + lq * APL + (g * 16 + li) * AS : a_tile + (g * 16 + li) * AS + 4 * lq;
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp)
      a[sp] = (DBG & 8) ? floatx4{(float)g, 1.f, 2.f, (float)sp} : *(const floatx4 *)(ap + sp * (QP ? 4 : 16));
  };
  auto mma_group = [&](const floatx4 (&a)[NS16], floatx4 (&acc)[COS]) __attribute__((always_inline)) {
    // one accumulator chain per slice: with two waves per SIMD the 40-cycle dependent latency is
    // covered by the partner wave (two chains + a final add measured 3-5 % slower)
#pragma unroll
    for (int q = 0; q < COS; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp)
#pragma unroll
      for (int q = 0; q < COS; ++q) {
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].x, a[sp].x, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].y, a[sp].y, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].z, a[sp].z, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].w, a[sp].w, acc[q], 0, 0, 0);
      }
  };
  auto mma_group2 = [&](const floatx4 (&a0)[NS16], const floatx4 (&a1)[NS16], floatx4 (&acc0)[COS],
                        floatx4 (&acc1)[COS]) __attribute__((always_inline)) {
    // two pair groups at once: two independent accumulator chains, MFMAs alternate between them
#pragma unroll
    for (int q = 0; q < COS; ++q) {
      acc0[q] = floatx4{0.f, 0.f, 0.f, 0.f};
      acc1[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp)
#pragma unroll
      for (int q = 0; q < COS; ++q) {
        acc0[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].x, a0[sp].x, acc0[q], 0, 0, 0);
        acc1[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].x, a1[sp].x, acc1[q], 0, 0, 0);
        acc0[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].y, a0[sp].y, acc0[q], 0, 0, 0);
        acc1[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].y, a1[sp].y, acc1[q], 0, 0, 0);
        acc0[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].z, a0[sp].z, acc0[q], 0, 0, 0);
        acc1[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].z, a1[sp].z, acc1[q], 0, 0, 0);
        acc0[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].w, a0[sp].w, acc0[q], 0, 0, 0);
        acc1[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].w, a1[sp].w, acc1[q], 0, 0, 0);
      }
  };
  auto scatter = [&](int orow, const floatx4 (&acc)[COS]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < COS; ++q) {
      if (DBG & 1) {
        asm volatile("" ::"v"(acc[q]));
      } else {
        floatx4 *op = (floatx4 *)(o_tile + orow + (w * COS + q) * 16 + 4 * lq);
        *op += acc[q];  // plain read-add-write: ds_add_f32 measured 2.5x slower end to end
      }
    }
  };

  if (nphase > 0) {
    int2 pr[NIDX];
    // Prefetch distances: gathered rows and weight fragments ONE phase ahead (two phases ahead with a
    // second register set measured 3-4 % slower: the kernel is not waiting on that latency), pair
    // indices FOUR lists ahead (requested during list kk, parked in registers over list kk+1's first
    // phase, written to one of 4 LDS buffers at its end: the list that used the buffer is done).
    load_indices(0, pr);
    store_indices(0, pr);
    if (nk > 1) {
      load_indices(1, pr);
      store_indices(1, pr);
    }
    if (nk > 2) {
      load_indices(2, pr);
      store_indices(2, pr);
    }
    int2 prq[NIDX];  // indices of a later list, in flight / parked
    int prq_list = -1;
    if (nk > 3) {
      load_indices(3, prq);
      prq_list = 3;
    }
    __syncthreads();
    issue_rows(0, preA);
    issue_weights(0, wcur);

    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    auto phase = [&](int p, floatx4 (&pre)[NSLOT]) __attribute__((always_inline)) {
      const int kk = p / nch, chn = p - kk * nch;
      const int ngroups = (uni(kseg[kk][1]) + 15) >> 4;
      const bool more = p + 1 < nphase;
      if (DBG & 256) t0 = stamp();
      store_rows(p, pre);  // waits for the row loads issued one phase ago
      if (DBG & 256) { t1 = stamp(); seg[0] += t1 - t0; t0 = t1; }
      const bool fetch_idx = (chn == 0) && (kk + 4 < nk);
      const bool park_idx = (chn == 0) && prq_list == kk + 3;  // indices requested one list ago
      int2 prn[NIDX];
      if (DBG & 512) __syncthreads(); else lds_barrier();
      if (DBG & 256) { t1 = stamp(); seg[1] += t1 - t0; t0 = t1; }
      // Next phase's loads, issued behind the barrier (in front of it they queue behind the staging writes
      // in the LDS pipe and measured 2.5 % slower).  Program order = vmcnt order: the index load and the
      // weight fragments are consumed at the end of this phase, the gathered rows only at the start of
      // the next one -> request the rows LAST, so those waits leave the 8 row loads in flight.
      if (fetch_idx) load_indices(kk + 4, prn);
      if (more) issue_weights(p + 1, wnext);
      if (more) issue_rows(p + 1, pre);  // this register set was just written to LDS
      if (DBG & 256) { t1 = stamp(); seg[2] += t1 - t0; t0 = t1; }

      // software pipeline over the pair groups: fragments of group g+1 are read while the MFMAs of
      // group g run; the result of group g is added to the output tile behind the MFMAs of group g+1
      const int *olb = olbuf[kk & 3];
      if (!(DBG & 32) && NWM == 1) {
        // NG pair groups per pass: NG independent accumulator chains whose MFMAs alternate, all A
        // fragments requested up front, all NG results added to the output tile behind the MFMAs
        auto pass = [&](int g, auto ng_tag) __attribute__((always_inline)) {
          constexpr int NG = decltype(ng_tag)::value;
          floatx4 a[NG][NS16], c[NG][COS];
          int orow[NG];
#pragma unroll
          for (int i = 0; i < NG; ++i) {
            orow[i] = olb[(g + i) * 16 + li] * OS;
            load_a(g + i, a[i]);
#pragma unroll
            for (int q = 0; q < COS; ++q) c[i][q] = floatx4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int sp = 0; sp < NS16; ++sp)
#pragma unroll
            for (int q = 0; q < COS; ++q) {
#pragma unroll
              for (int i = 0; i < NG; ++i)
                c[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].x, a[i][sp].x, c[i][q], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NG; ++i)
                c[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].y, a[i][sp].y, c[i][q], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NG; ++i)
                c[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].z, a[i][sp].z, c[i][q], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < NG; ++i)
                c[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[q][sp].w, a[i][sp].w, c[i][q], 0, 0, 0);
            }
#pragma unroll
          for (int i = 0; i < NG; ++i) scatter(orow[i], c[i]);
        };
        int g = 0;
        if (!(DBG & 64))
          for (; g + 3 < ngroups; g += 4) pass(g, std::integral_constant<int, 4>{});
        for (; g + 1 < ngroups; g += 2) pass(g, std::integral_constant<int, 2>{});
        if (g < ngroups) pass(g, std::integral_constant<int, 1>{});
      } else {
      floatx4 a_cur[NS16], a_nxt[NS16], acc_cur[COS], acc_prev[COS];
      int orow_prev = 0;
      if (wm < ngroups) {
        load_a(wm, a_cur);
        for (int g = wm; g < ngroups; g += NWM) {
          const int orow = olb[g * 16 + li] * OS;
          if (g + NWM < ngroups) load_a(g + NWM, a_nxt);
          mma_group(a_cur, acc_cur);
          if (g > wm) scatter(orow_prev, acc_prev);
#pragma unroll
          for (int q = 0; q < COS; ++q) acc_prev[q] = acc_cur[q];
          orow_prev = orow;
#pragma unroll
          for (int sp = 0; sp < NS16; ++sp) a_cur[sp] = a_nxt[sp];
        }
        scatter(orow_prev, acc_prev);
      }
      }
      if (DBG & 256) { t1 = stamp(); seg[3] += t1 - t0; t0 = t1; }
      if (park_idx) store_indices(kk + 3, prq);
      if (fetch_idx) {
#pragma unroll
        for (int j = 0; j < NIDX; ++j) prq[j] = prn[j];
        prq_list = kk + 4;
      }
      if (more) {
#pragma unroll
        for (int q = 0; q < COS; ++q)
#pragma unroll
          for (int sp = 0; sp < NS16; ++sp) wcur[q][sp] = wnext[q][sp];
      }
      if (DBG & 256) { t1 = stamp(); seg[4] += t1 - t0; t0 = t1; }
      if (DBG & 512) __syncthreads(); else lds_barrier();  // a_tile and this list's ol[] are free again
      if (DBG & 256) { t1 = stamp(); seg[5] += t1 - t0; }
    };
    for (int p = 0; p < nphase; ++p) phase(p, preA);
    if ((DBG & 256) && lane == 0)
      for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[i], seg[i]);
  }
  const int rows = (int)min((int64_t)M, n_out - o0);
  // ksplit > 1: `out` is the launcher's workspace [ksplit][n_out][Co]; this workgroup's partial tile (zeros when it
  // owns no offset here) goes to slab blockIdx.y and k_split_reduce adds the slabs in a FIXED order: no float
  // atomics, the result does not depend on which workgroup finishes first.
  float *dst = out + (int64_t)blockIdx.y * n_out * Co;
  bool finished = ksplit == 1;  // this workgroup holds the finished tile in LDS (statistics below)
  if (ksplit > 1 && tickets != nullptr) {
    // ... and (round 4) the LAST of a tile's ksplit workgroups to get here does that addition itself, in the same fixed order,
    // instead of a second launch: 26 launches of k_split_reduce per train step and their gaps are gone, and the complete tile
    // is in LDS once more, so the column statistics / batch-norm-backward sums below are taken from it as in an unsplit launch
    // (the coarse levels' batch norms had to make their own statistics pass).  The hand-off that the L2 caches of eight XCDs
    // require: write-through (sc1) stores of the partial tile, every wave drains them, the barrier, ONE lane draws the tile's
    // ticket; the workgroup that draws the last one reads all slabs with sc1 loads.  The ticket word goes back to 0.
    const size_t slab_bytes = (size_t)n_out * (size_t)Co * 4;
    const unsigned row_b = (unsigned)Co * 4u;
    const unsigned tile_b = (unsigned)(((size_t)o0 * (size_t)Co + (size_t)co0) * 4);  // (< 4 GiB: the slabs of a split launch are a few MB)
    {
      const __amdgpu_buffer_rsrc_t ds = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, (int)slab_bytes, 0x00020000);
      for (int i = tid; i < rows * (COB / 4); i += NT) {
        const int r = i / (COB / 4), c4 = i % (COB / 4);
        const floatx4 v = *(const floatx4 *)(o_tile + r * OS + c4 * 4);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, ds,
                                               tile_b + (unsigned)r * row_b + (unsigned)c4 * 16u, 0, 16 /* sc1 */);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) nk_s = (int)__hip_atomic_fetch_add(tickets + lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(nk_s) != ksplit - 1) return;
    if (tid == 0) __hip_atomic_store(tickets + lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");
    // (slab 0 first -- it holds bias and shortcut --, then 1, 2, ...: k_split_reduce's order.  The split is 3-, 9- or 27-fold (8-fold
    //  for a strided rulebook): three slabs' worth of loads are in flight at a time, eight 16-byte pieces per thread each)
    constexpr int NE = M * (COB / 4) / NT;
    floatx4 acc[NE];
    unsigned off[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = tid + e * NT, r = i / (COB / 4), c4 = i % (COB / 4);
      off[e] = r < rows ? tile_b + (unsigned)r * row_b + (unsigned)c4 * 16u : 0x80000000u;  // (out of range: reads zeros)
    }
    for (int j0 = 0; j0 < ksplit; j0 += 3) {
      u32x4 b[3][NE];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const __amdgpu_buffer_rsrc_t ps = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)min(j0 + u, ksplit - 1) * (size_t)n_out * (size_t)Co), 0,
                                                                            (int)slab_bytes, 0x00020000);
#pragma unroll
        for (int e = 0; e < NE; ++e) b[u][e] = __builtin_amdgcn_raw_buffer_load_b128(ps, off[e], 0, 16 /* sc1 */);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (j0 + u < ksplit) {
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const floatx4 bf = floatx4{__uint_as_float(b[u][e].x), __uint_as_float(b[u][e].y), __uint_as_float(b[u][e].z), __uint_as_float(b[u][e].w)};
            acc[e] = (j0 + u == 0) ? bf : acc[e] + bf;
          }
        }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = tid + e * NT, r = i / (COB / 4), c4 = i % (COB / 4);
      if (r < rows) {
        *(floatx4 *)(o_tile + r * OS + c4 * 4) = acc[e];
        *(floatx4 *)(final_out + (o0 + r) * Co + co0 + c4 * 4) = acc[e];
      }
    }
    __syncthreads();
    finished = true;
  } else
  for (int i = tid; i < rows * (COB / 4); i += NT) {
    const int r = i / (COB / 4), c4 = i % (COB / 4);
    *(float4 *)(dst + (o0 + r) * Co + co0 + c4 * 4) = *(const float4 *)(o_tile + r * OS + c4 * 4);
  }
  // Column statistics of the finished tile (sum, sum of squares per output plane over the tile's rows) while it is still
  // in LDS: the batch norm that consumes this tensor takes them instead of reading the tensor once more
  // (colstats [ntiles][2][Co], one deterministic partial per tile; the fp64 finalize of scn_bn_forward adds them up).
  if (colstats.out != nullptr && finished) {
    constexpr int G = NT / COB;  // threads per column
    const int c = tid % COB, g = tid / COB;
    float cs = 0.f, css = 0.f;
    if (colstats.x == nullptr) {
      for (int r = g; r < rows; r += G) {
        const float v = o_tile[r * OS + c];
        cs += v;
        css += v * v;
      }
    } else {  // the two reductions of the backward pass of the batch norm over x whose output this launch differentiates
      const float m = colstats.mean[co0 + c];
      const float wv = colstats.invstd[co0 + c] * (colstats.gamma ? colstats.gamma[co0 + c] : 1.f);
      const float bv = fmaf(-m, wv, colstats.beta ? colstats.beta[co0 + c] : 0.f);
      const float leak = colstats.leak;
      for (int r = g; r < rows; r += G) {
        const float xv = colstats.x[(o0 + r) * Co + co0 + c];
        float d = o_tile[r * OS + c];
        d = (leak == 1.f || fmaf(xv, wv, bv) > 0.f) ? d : d * leak;
        cs += d;
        css += (xv - m) * d;
      }
    }
    float *red = a_tile;  // free: every wave is past the last phase's closing barrier (or no phase ran)
    red[(g * COB + c) * 2 + 0] = cs;
    red[(g * COB + c) * 2 + 1] = css;
    __syncthreads();
    if (tid < COB) {
      cs = 0.f; css = 0.f;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        cs += red[(j * COB + tid) * 2 + 0];
        css += red[(j * COB + tid) * 2 + 1];
      }
      colstats.out[((int64_t)tile * 2 + 0) * Co + co0 + tid] = cs;
      colstats.out[((int64_t)tile * 2 + 1) * Co + co0 + tid] = css;
    }
  }
}

// out[i] = part[0][i] + part[1][i] + ... + part[ksplit-1][i], always in this order
__global__ void __launch_bounds__(256)
    k_split_reduce(const float *__restrict__ part, float *__restrict__ out, int64_t total4, int ksplit) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    float4 a = ((const float4 *)part)[i];
    for (int j = 1; j < ksplit; ++j) {
      const float4 b = ((const float4 *)part)[(int64_t)j * total4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    ((float4 *)out)[i] = a;
  }
}

// ------------------------------------ forward, barrier-free wave-private (v4) ---
// Same tile (128 output rows x 64 planes kept in LDS) as k_conv_tile2, different division of labour: a workgroup
// is TWO waves, each owning 32 output planes (two 16-plane slices) and walking the offsets on its own.  The MFMA B
// operand (gathered rows) is loaded straight from global memory in fragment shape -- lane (pair li, k-quad lq)
// fetches 16 bytes of ITS pair's row -- so there is no LDS staging tile, no ds_write, and, because a wave only ever
// touches its own planes of the output tile, NO barrier after the tile set-up.  Price: every row is fetched by both
// waves (64-byte segments).
//
// The unit of work is a PASS: up to two 16-pair groups of one (offset, C_i chunk) phase, 32 or 64 MFMAs per wave.  All
// passes of the tile are tabulated in LDS during set-up (8 bytes each), so the loop itself carries almost no scalar
// bookkeeping (the first version generated the pass sequence on the fly: ~1100 of ~4800 cycles per pass).  Everything
// that is not an MFMA is issued in "slots" between groups of four MFMAs, so that a wave stalled on the vector-memory
// issue path still has MFMAs queued:
//   pass p computes | its rows and weights are in registers | rows of pass p+1 were requested during pass p-1 |
//   during pass p: row offsets of pass p+2 from its (arrived) indices, weight loads of pass p+1, index loads of
//   pass p+3, row loads of pass p+2, read-add-write of pass p-1's result into the output tile, table entry p+4.
// The loads are inline asm with hand-counted s_waitcnt: hipcc's own counters collapse to vmcnt(0..8) at the top of
// every pass once loads stay in flight over the back edge of the unrolled loop.  Every pass issues the same loads in
// the same order (8 weight, 2 index, 8 row); the ones a pass does not need read the first bytes of their buffer.  hipcc does not know that an asm load's destination is not valid
// until the wait, so: the waits name the destinations as INPUTS (keeps them allocated; a "+v" tie made hipcc copy
// them in FRONT of the wait), nothing but row registers (fixed variables of a 3-way unrolled loop) is in flight over
// the back edge, and tools/check_asm_loads.py (run by build.py) proves on the emitted assembly that no instruction
// touches a register with a load in flight.
// GT = 1: the pass table lives in GLOBAL memory (a 64-byte aligned region per workgroup of the launcher's workspace) and is
// read with scalar loads: an entry in LDS costs a v_mov (address) + ds_read + two v_readfirstlane per pass -- three
// vector-ALU instructions at 8-12 matrix-pipe cycles each -- a scalar load none.
// SK = 1 (needs GT = 1): STREAM-K (DESIGN.md section 3.3e has the whole story and every measurement).  The launch is 1024
// persistent workgroups (four per CU) and what a workgroup is given is not just tiles: the passes of all (tile, column block) items
// form one sequence (`tile_passes`: the prefix of L = passes per tile and input chunk over the tiles, scn_tile_passes); all but the
// last 1 + fraction rounds of 1024 items are walked WHOLE, the rest of the sequence is cut into 1024 slices.  Whole tiles dealt
// over 1,024 resident workgroups run ceil(n / 1024) rounds (2,766 tiles: three rounds for 2.70 rounds of work; 4,688: five for
// 4.58).  A slice begins and ends anywhere inside a tile's pass table, so an item cut by a slice boundary is computed in two PIECES
// that are chained, not summed: the piece that holds the item's first passes stores its 128 x 64 output tile to a slab
// (write-through, sc1), the piece that holds the rest starts its LDS tile from that slab and goes on -- every output element
// still sees its contributions in (offset, chunk, pair) order, so the result is bit-identical to the one-workgroup-per-tile
// launch and to k_conv_tile2.  Within a slice the piece that FEEDS the next slice runs first and the piece that WAITS for the
// previous slice last (a slice is >= 1 tile of work: the slab it waits for was published a tile ago).  Two schedules: STATIC
// (sk_static, the default: nothing is drawn, see sk_next) and DRAWN (items and slices from per-XCD counters, chunks of SK_CH
// consecutive slices dealt round robin over the XCDs; a slice waits only for a lower one that was drawn earlier or belongs to
// another XCD's sequence and the lowest unfinished slice never waits: no deadlock whatever the residency).  `sk_state` (zero when
// the launch starts and when it ends): [0..7] whole-item counters, [8] exit ticket, [9] set if a wait for a slab gave up (never,
// unless the launch is broken), [10] slices drawn so far, [16..23] slice counters, [32 + s] "slab s is published".
constexpr int SK_CH = SCN_SK_CH;    // slices per chunk of one XCD (16: the halo rows of a 14-tile region were fetched by two XCDs)
constexpr int SK_ITEM_CH = 64;     // whole items per chunk of one XCD (the first rounds)
constexpr int SK_STATE_WORDS = 32; // counters + ticket in front of the flags
constexpr bool SK_DRAIN = true;
constexpr bool SK_PREFETCH_DRAW = false;
constexpr int SK_SLICES = SCN_SK_SLICES;  // slices of the sliced part of a launch (= the entries of scn_tile_passes' slice tables)
// SPLIT = 1 (needs GT = 1, not SK): a launch that splits its offsets over ksplit workgroups per (tile, column block) -- the coarse
// UNet levels, a few hundred to a few thousand sites -- adds its partial tiles ITSELF (round 5; k_conv_tile2 does since round 4):
// every workgroup stores its partial tile to slab blockIdx.y of `out_a` (write-through), draws the item's ticket
// (`sk_state_a` = the ticket words), and the last one to arrive reads the ksplit slabs back in slab order, writes the sum to
// `slabs_a` (= the launch's real output) and emits the column statistics.  The pieces of a tile are ADDED by its last arriver, not
// chained as stream-K's are: the ksplit workgroups of a tile run at the same time.  Same slabs, same order as k_conv_tile2's
// split launch: bit-identical to it.
// ASM = 1 (needs GT = 1, INCOL >= 0, DBG = 0): the pass loop is the hand-written one of tile4_loop_asm.inc (tools/gen_tile4_loop.py:
// six copies of the pass body -- three rotations of the row sets x two roles of the weight sets --, 12 vector-ALU instructions
// per pass instead of ~26, lanes past the end of a list masked out of the loads by EXEC masks from the table) over 16-byte table
// entries; same arithmetic in the same order, bit-identical to the C++ loop.
template <int DBG = 0, int INCOL = -1, int GT = 0, int SK = 0, int SPLIT = 0, int ASM = 0>  // INCOL: -1 = read the argument, 0 / 1 = known at compile time (no v_cndmask per pair)
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))  // (<= 256 registers: hipcc then keeps the accumulators in VGPRs, no v_accvgpr_read)
    k_conv_tile4(const float *__restrict__ in, float *__restrict__ out_a, const float *__restrict__ Wp,
                 const float *__restrict__ bias_a, const float *__restrict__ residual_a, StatsArg colstats_a, const int2 *__restrict__ rules,
                 const int32_t *__restrict__ bases_a, int K_a, int64_t stride_a, int in_col,
                 int64_t n_out_a, int n_gran_a, int Ci, int Co, int ntiles_a, int ncb_a,
                 unsigned in_bytes, unsigned wp_bytes, int ksplit_a,
                 unsigned long long *__restrict__ gtab_a = nullptr, int gstride_a = 0,
                 const int32_t *__restrict__ tile_passes_a = nullptr, unsigned *sk_state_a = nullptr, float *slabs_a = nullptr,
                 int na_items_a = 0, int sk_static_a = 0) {
  // (mutable copies of everything the set-up and the epilogue of a piece use: a stream-K launch parks them in the lanes of
  //  one VGPR while the pass loop runs, see `park_all` below)
  float *out = out_a;
  const float *bias = bias_a, *residual = residual_a;
  StatsArg colstats = colstats_a;
  const int32_t *bases = bases_a, *tile_passes = tile_passes_a;
  int K = K_a, n_gran = n_gran_a, ntiles = ntiles_a, ncb = ncb_a, gstride = gstride_a, na_items = na_items_a < 0 ? -na_items_a : na_items_a;
  const int ksplit = SK ? 1 : ksplit_a;
  int64_t stride = stride_a, n_out = n_out_a;
  unsigned long long *gtab = gtab_a;
  unsigned *sk_state = sk_state_a;
  float *slabs = slabs_a;
  constexpr int M = 128, COB = 64, OS = COB + 4, CK = 64, NS16 = 4, COS = 2;
  constexpr int TMAX = KMAX * 4 * SCN_TILE4_MAX_CHUNKS + 8;  // passes per tile: <= 4 per (offset, chunk); + sentinels
  __shared__ __attribute__((aligned(16))) float o_tile[(M + 1) * OS];  // row M: sink for padded pairs
  __shared__ __attribute__((aligned(8))) unsigned ptab[GT ? 1 : TMAX][2];  // {byte offset of the pass's first pair in `rules`,
                                                                  //  pairs left in the list | chunk<<8 | k<<12 | first<<20}
  static_assert(!SK || GT, "stream-K keeps its pass tables in global memory");
  static_assert(!ASM || (GT && INCOL >= 0 && DBG == 0), "the assembly loop reads its pass table with scalar loads and knows the rulebook column");
  static_assert(!SPLIT || (GT && !SK), "the fused split launch keeps its pass tables in global memory and is not a stream-K launch");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const bool icol = INCOL < 0 ? in_col != 0 : INCOL != 0;
  // a * b + c in ONE full-rate instruction (a, b < 2^24); hipcc turns the C expression into v_sub + v_mul_lo_u32 (quarter rate)
  auto mad24 = [](unsigned a_, unsigned b_, unsigned c_) __attribute__((always_inline)) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a_), "s"(b_), "v"(c_));
    return r;
  };
  const unsigned row_bytes = (unsigned)((colstats.ldi ? colstats.ldi : Ci) * 4), os_u = (unsigned)(OS * 4);
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Wp, 0, (int)wp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ru_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)rules, 0, (int)(K * stride * 8), 0x00020000);

  const int nch = Ci / CK;
  auto uni = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  // (a pointer that IS uniform, said so: hipcc otherwise wraps buffer accesses through a resource made from it in a waterfall loop)
  auto uni_ptr = [](const void *q) __attribute__((always_inline)) {
    const unsigned long long a = (unsigned long long)(size_t)q;
    return (void *)(size_t)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32);
  };
  // Wave shuffles with the lane id handed in.  HIP's __shfl* take it from v_mbcnt, which hipcc computes once at the top of the
  // kernel and keeps: in the ASM variant that register would live across the assembly pass loop (spilled to scratch, and see the
  // note at the loop about vector values that cross it) -- there the caller passes the lane id it took from the hardware last.
  auto shfl_i = [](int v, int src_lane) __attribute__((always_inline)) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); };
  auto shfl_f = [](float v, int src_lane) __attribute__((always_inline)) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
  };
  // ---- what this workgroup computes: ONE item = (tile, column block), all of its passes -- or (SK) a sequence of PIECES:
  // passes [p_lo, p_hi) of an item's table, starting from bias + residual or from slab `seed_slab`, ending in `out` (with the
  // column statistics) or in slab `out_slab`
  int item = 0, p_lo = 0, p_hi = 0x7fffffff, seed_slab = -1, out_slab = -1;
  if (!SK) {
    // Workgroup b runs on XCD b % 8.  Each XCD takes CHUNKS of consecutive tiles (neighbouring tiles gather
    // overlapping rows: shared L2), dealt round robin so that density differences along the row order (whole scenes)
    // do not leave some XCDs with more work than others; the ragged end falls back to one contiguous range each.
    const int total = ntiles * ncb;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int CH = 64 * ncb;
    const int full = (DBG & 512) ? 0 : (total / (8 * CH)) * (8 * CH);  // part that divides into rounds of 8 chunks
    if (bid < full) {
      item = ((j / CH) * 8 + xcd) * CH + j % CH;
    } else {
      const int rest = total - full, b2 = bid - full, x2 = b2 & 7, j2 = b2 >> 3;
      const int q = rest >> 3, r = rest & 7;
      item = full + (x2 < r ? x2 * (q + 1) : r * (q + 1) + (x2 - r) * q) + j2;
    }
  }
  // ---- stream-K bookkeeping (scalar; both waves compute the same values)
  int sk_phase = 0, sk_par = 0, sk_steal = 0, sk_steal_a = 0, sk_s = 0, sk_ia = 0, sk_pa = 0, sk_ib = 0, sk_pb = 0, sk_cur = 0, gpar = 0,
      sk_pf = 0, sk_cb = 0, sk_static = sk_static_a;  // (sk_ia / sk_ib / sk_cur: TILES; sk_cb: the slice's column block)
#ifdef SCN_TUNING
  if (na_items_a < 0) sk_steal = 8;  // (A/B: whole items only, no sliced part; the item count must be a multiple of SCN_SK_SLICES / 2)
#endif
  unsigned *sk_flags = sk_state + SK_STATE_WORDS;
  // ---- SK: the scalars that only the set-up and the epilogue of a piece need -- kernel arguments and the slice bookkeeping,
  // some fifty of them -- would all be live across the pass loop of a persistent kernel, and hipcc, short of scalar registers,
  // spills by loop DEPTH: the search loops of the set-up are nested deeper than the pass loop, so it was the pass loop's scalars
  // that went to VGPR lanes (27 v_readlane per three passes, each 8-12 matrix-pipe cycles).  They are parked by hand instead:
  // lane i of ONE VGPR holds value i from just before the pass loop to just behind it, where everything is read back.
  int park = 0;
  int tile_parked = 0;
  auto park_all = [&](bool store) __attribute__((always_inline)) {
    // (the lane is an immediate: v_writelane takes ONE scalar register operand)
#define SCN_P32(I, V)                                                                                                      \
  do {                                                                                                                     \
    if (store) asm volatile("v_writelane_b32 %0, %1, " #I : "+v"(park) : "s"(__builtin_amdgcn_readfirstlane((int)(V)))); \
    else V = (decltype(V))__builtin_amdgcn_readlane(park, I);                                                              \
  } while (0)
#define SCN_P64(I, J, V)                                                                                                   \
  do {                                                                                                                     \
    unsigned long long v64_ = (unsigned long long)(V);                                                                     \
    int lo_ = (int)(unsigned)v64_, hi_ = (int)(unsigned)(v64_ >> 32);                                                      \
    SCN_P32(I, lo_);                                                                                                       \
    SCN_P32(J, hi_);                                                                                                       \
    if (!store) V = (decltype(V))((unsigned long long)(unsigned)lo_ | (unsigned long long)(unsigned)hi_ << 32);            \
  } while (0)
    if (!store) asm volatile("" : "+v"(park));
    SCN_P64(0, 1, out); SCN_P64(2, 3, bias); SCN_P64(4, 5, residual);
    SCN_P64(6, 7, colstats.out); SCN_P64(8, 9, colstats.x); SCN_P64(10, 11, colstats.mean); SCN_P64(12, 13, colstats.invstd);
    SCN_P64(14, 15, colstats.gamma); SCN_P64(16, 17, colstats.beta);
    int leak_bits = (int)__float_as_uint(colstats.leak);
    SCN_P32(18, leak_bits);
    if (!store) colstats.leak = __uint_as_float((unsigned)leak_bits);
    SCN_P64(19, 20, bases); SCN_P64(21, 22, tile_passes); SCN_P64(23, 24, gtab); SCN_P64(25, 26, sk_state); SCN_P64(27, 28, slabs);
    SCN_P32(29, K); SCN_P32(30, n_gran); SCN_P32(31, ntiles); SCN_P32(32, ncb); SCN_P32(33, gstride); SCN_P32(34, sk_pf);
    SCN_P64(36, 37, stride); SCN_P64(38, 39, n_out);
    SCN_P32(40, sk_phase); SCN_P32(41, sk_par); SCN_P32(42, sk_steal); SCN_P32(43, sk_s); SCN_P32(44, sk_ia); SCN_P32(45, sk_pa);
    SCN_P32(46, sk_ib); SCN_P32(47, sk_pb); SCN_P32(48, sk_cur); SCN_P32(49, gpar);
    SCN_P32(50, item); SCN_P32(51, out_slab); SCN_P32(52, tile_parked); SCN_P32(53, sk_steal_a); SCN_P32(54, na_items); SCN_P32(55, sk_cb); SCN_P32(56, sk_static);
#undef SCN_P64
#undef SCN_P32
    if (store) asm volatile("" : "+v"(park));
    else sk_flags = sk_state + SK_STATE_WORDS;
  };
  // LDS words for the workgroup's two waves to agree on what was drawn: the row behind the output tile (the sink row of older
  // builds), two alternating sets
  int *const sk_slot = (int *)(o_tile + M * OS);
  // where slice s begins: entry s of the slice table scn_tile_passes left behind the pass prefix for this column-block count
  // (item, pass inside the item's table in units of nch passes).  One scalar load for both ends of a slice.
  auto sk_bounds = [&](int s_) __attribute__((always_inline)) {
    // slice s_ = (tile slice s_ / ncb, column block s_ % ncb): the sliced part of the TILE sequence is cut into 1024 / ncb equal
    // parts and each part is walked once per column block, by ncb slices that are drawn one after the other -- so the column
    // blocks of a tile, which gather the same rows, are in flight on one XCD at the same time (walked one after the other by
    // one workgroup, the second walk came an item's time later and missed the L2: 2.5 x the fabric reads of the plain launch)
    // (table slots: ncb - 1 for the drawn schedule, and 2 / 4 / 5 for the static schedule with 1 / 2 / 4 column blocks)
    const int slot = sk_static ? (ncb == 1 ? 2 : ncb == 2 ? 4 : 5) : ncb - 1;
    const int2 *e = (const int2 *)(tile_passes + ((ntiles + 2) & ~1)) + (size_t)slot * (SK_SLICES + 1) + s_ / ncb;
    const int2 a = e[0], b = e[1];
    sk_ia = uni(a.x); sk_pa = uni(a.y) * nch;
    sk_ib = uni(b.x); sk_pb = uni(b.y) * nch;
    sk_cb = s_ % ncb;
  };
  // One draw from this XCD's whole-item counter (which = 1: the first rounds of the launch deal WHOLE items, in the order the
  // one-workgroup-per-tile launch starts them -- chunks of consecutive items per XCD: no slab, no table) or from its slice
  // counter (which = 2).  Almost every draw was made ahead of time, while the previous piece's first loads were in flight
  // (sk_pf: see the prologue of the pass loop), and all that is left of it here is a barrier: a draw made HERE is a round
  // trip through a memory path that a CU's eight gathering waves keep full, 2-3 us -- the first version also tried the other
  // seven XCDs' counters when its own ran out, fifteen such round trips at the end of every workgroup: 8 % of a launch.
  // (Every XCD holds the same number of workgroups and is dealt the same number of items and equal slices: nothing to steal.)
  // "this is thread 0 of the workgroup", without a vector register that lives across the assembly pass loop (see there: a copy of
  // such a register may be made while wave 1 is masked off)
  auto thread0 = [&]() __attribute__((always_inline)) {
    if (!ASM) return tid == 0;
    int l_;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
    return w == 0 && l_ == 0;
  };
  auto sk_draw = [&](int which) __attribute__((always_inline)) {
    if (sk_pf == which) {
      sk_pf = 0;
      lds_barrier();
    } else {
      sk_pf = 0;  // (a draw of the other kind made ahead is dropped: that counter has run out)
      if (thread0())
        sk_slot[sk_par] = (int)__hip_atomic_fetch_add(sk_state + (which == 1 ? 0 : 16) + ((int)blockIdx.x & 7), 1u, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
      lds_barrier();
    }
    const int t = uni(sk_slot[sk_par]);
    sk_par ^= 1;
    return t;
  };
  // the next piece: of the current slice -- the item its END cuts first (that piece feeds the next slice), the whole items,
  // the item its BEGINNING cuts last (that piece waits for the previous slice) -- or of a new slice
  auto sk_next = [&]() __attribute__((always_inline)) {
#ifdef SCN_TUNING
    if (na_items_a == (int)0x80000000) {  // (A/B: this binary, but ONE whole item per workgroup as the hardware deals them)
      if (sk_cur) return false;
      sk_cur = 1;
      const int total = ntiles * ncb, bid = blockIdx.x, xcd = bid & 7, j = bid >> 3, CH = 64 * ncb;
      const int full = (total / (8 * CH)) * (8 * CH);
      if (bid < full) item = ((j / CH) * 8 + xcd) * CH + j % CH;
      else {
        const int rest = total - full, b2 = bid - full, x2 = b2 & 7, j2 = b2 >> 3, q = rest >> 3, r = rest & 7;
        item = full + (x2 < r ? x2 * (q + 1) : r * (q + 1) + (x2 - r) * q) + j2;
      }
      p_lo = 0; p_hi = 0x7fffffff; seed_slab = -1; out_slab = -1;
      return true;
    }
#endif
    if (sk_static) {
      // STATIC schedule: workgroup (XCD x = b % 8, j = b / 8) walks item (r * 8 + x) * 128 + j in round r of the whole items, then
      // slice (c * 8 + x) * SK_CH + i with j = c * SK_CH + i -- nothing is drawn, no counter, no barrier for it.  Workgroups j
      // and j + 1 of an XCD walk the column blocks of the SAME tiles for the whole launch: identical work, so they stay in
      // step and one fetch of a gathered row from the fabric serves both (with draws, whoever finished next took the next
      // item: the pairs drifted apart and the launch read 1.7-2.5 x the plain launch's bytes through the fabric).  The slice
      // table is built for this schedule: a slice is as much shorter as its owner's whole items are heavier than average
      // (scn_tile_passes), which is all the balancing a dynamic draw did on an undisturbed chip.
      for (;;) {
        if (sk_phase == 0) {
          if (sk_cur < (na_items >> 10)) {  // rounds of whole items
            item = ((sk_cur++ * 8) + ((int)blockIdx.x & 7)) * 128 + ((int)blockIdx.x >> 3);
            p_lo = 0; p_hi = 0x7fffffff; seed_slab = -1; out_slab = -1;
            return true;
          }
          if (sk_steal) return false;  // (the slice is done)
          sk_steal = 1;
          {
            const int j = (int)blockIdx.x >> 3;
            sk_s = ((j / SK_CH) * 8 + ((int)blockIdx.x & 7)) * SK_CH + j % SK_CH;
          }
          sk_bounds(sk_s);
          if (sk_ia == sk_ib) {
            // (an EMPTY slice -- both ends at one point -- inside a tile still hands the tile on: a piece of no passes that
            //  starts from the previous slab and publishes its own; at a tile's first pass there is nothing to hand on)
            if (sk_pa == sk_pb && sk_pa == 0) return false;
            item = sk_ia * ncb + sk_cb; p_lo = sk_pa; p_hi = sk_pb;
            seed_slab = sk_pa > 0 ? sk_s - ncb : -1;
            out_slab = sk_s;
            sk_phase = 4;
            return true;
          }
          sk_phase = 1;
        }
        if (sk_phase == 1) {
          sk_phase = 2;
          sk_cur = sk_pa > 0 ? sk_ia + 1 : sk_ia;
          if (sk_pb > 0) {
            item = sk_ib * ncb + sk_cb; p_lo = 0; p_hi = sk_pb; seed_slab = -1; out_slab = sk_s;
            return true;
          }
        }
        if (sk_phase == 2) {
          if (sk_cur < sk_ib) {
            item = sk_cur++ * ncb + sk_cb; p_lo = 0; p_hi = 0x7fffffff; seed_slab = -1; out_slab = -1;
            return true;
          }
          sk_phase = 3;
        }
        if (sk_phase == 3) {
          sk_phase = 4;
          if (sk_pa > 0) {
            item = sk_ia * ncb + sk_cb; p_lo = sk_pa; p_hi = 0x7fffffff; seed_slab = sk_s - ncb; out_slab = -1;
            return true;
          }
        }
        return false;
      }
    }
    for (;;) {
      if (sk_phase == 0) {
        if (!sk_steal_a) {
          const int t = sk_draw(1);
          if (t < (na_items >> 3)) {
            item = ((t / SK_ITEM_CH) * 8 + ((int)blockIdx.x & 7)) * SK_ITEM_CH + t % SK_ITEM_CH;
            p_lo = 0; p_hi = 0x7fffffff; seed_slab = -1; out_slab = -1;
            return true;
          }
          sk_steal_a = 1;  // (the whole items of this XCD are dealt)
        }
        if (sk_steal >= 8) return false;
        {
          // a slice of this XCD's sequence; when that has run out, of another XCD's -- workgroups enter the sliced part when their
          // last whole item ends, up to an item's time apart, and equal slices alone would carry that spread to the end of the
          // launch: the early ones take a second slice.  Word 10 counts the slices drawn by the whole launch, so that the
          // workgroups that finish last see with ONE load that nothing is left, not with seven more draws.
          int t = sk_steal == 0 ? sk_draw(2) : (SK_SLICES >> 3);
          while (t >= (SK_SLICES >> 3)) {
            if (sk_steal == 0) {
              if (thread0()) sk_slot[sk_par] = (int)__hip_atomic_load(sk_state + 10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              lds_barrier();
              const int taken = uni(sk_slot[sk_par]);
              sk_par ^= 1;
              if (taken >= SK_SLICES) { sk_steal = 8; return false; }
            }
            if (++sk_steal >= 8) return false;
            const int xc = ((int)blockIdx.x + sk_steal) & 7;
            if (thread0()) sk_slot[sk_par] = (int)__hip_atomic_fetch_add(sk_state + 16 + xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_barrier();
            t = uni(sk_slot[sk_par]);
            sk_par ^= 1;
          }
          if (thread0()) __hip_atomic_fetch_add(sk_state + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sk_s = ((t / SK_CH) * 8 + (((int)blockIdx.x + sk_steal) & 7)) * SK_CH + t % SK_CH;
        }
        sk_bounds(sk_s);
        if (sk_ia == sk_ib) {  // inside one item (a slice shorter than the item)
          if (sk_pa == sk_pb && sk_pa == 0) continue;  // (empty, at a tile's first pass: nothing to hand on; see the static branch)
          item = sk_ia * ncb + sk_cb; p_lo = sk_pa; p_hi = sk_pb;
          seed_slab = sk_pa > 0 ? sk_s - ncb : -1;
          out_slab = sk_s;
          return true;
        }
        sk_phase = 1;
      }
      if (sk_phase == 1) {
        sk_phase = 2;
        sk_cur = sk_pa > 0 ? sk_ia + 1 : sk_ia;
        if (sk_pb > 0) {
          item = sk_ib * ncb + sk_cb; p_lo = 0; p_hi = sk_pb; seed_slab = -1; out_slab = sk_s;
          return true;
        }
      }
      if (sk_phase == 2) {
        if (sk_cur < sk_ib) {
          item = sk_cur++ * ncb + sk_cb; p_lo = 0; p_hi = 0x7fffffff; seed_slab = -1; out_slab = -1;
          return true;
        }
        sk_phase = 3;
      }
      sk_phase = 0;
      if (sk_pa > 0) {
        item = sk_ia * ncb + sk_cb; p_lo = sk_pa; p_hi = 0x7fffffff; seed_slab = sk_s - ncb; out_slab = -1;
        return true;
      }
    }
  };
  unsigned long long seg[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
  int tid_opaque = (int)threadIdx.x;
  for (;;) {  // SK: pieces; otherwise once
  if (SK && !sk_next()) break;
  // (SK: everything per-lane below is derived afresh in every piece -- left loop-invariant, hipcc hoists a dozen per-lane
  //  addresses of the set-up and the epilogue out of the piece loop, finds no registers for them beside the pass loop's 220
  //  and spills them to scratch)
  if (SK && ASM) {  // (no vector register lives across the assembly pass loop: the thread id comes from the hardware every time)
    int l_;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
    tid_opaque = 64 * w + l_;
  } else if (SK) asm volatile("" : "+v"(tid_opaque));
  // SK: the two waves that share a SIMD take turns at the higher issue priority, piece by piece.  Waves that the hardware deals
  // come and go, so which of a SIMD's two is the older one (the one its arbiter prefers) alternates by itself; persistent waves
  // are all the same age for the whole launch, one of each pair is preferred throughout -- and the same pass loop ran 5 %
  // slower (4096 whole tiles on 1024 persistent workgroups: +7 % against hardware dispatch without this, +1.4 % with it;
  // profiles/r04_ab_stream_k.txt).  Which workgroups share SIMDs: those 2 x 256 apart in launch order (four per CU).
  if (SK && (!SK_TUNING_PRIO || g_sk_prio_dev)) {
    if ((((int)blockIdx.x >> 9) + gpar) & 1) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
  }
  // (not const: the ASM variant derives them afresh behind its pass loop, see there)
  int tid = SK ? tid_opaque : (int)threadIdx.x;
  int lane = tid & 63;
  const int li = lane & 15, lq = lane >> 4;
  const unsigned lq16 = (unsigned)(lq * 16);
  const int tile = item / ncb;
  const int co0 = (item % ncb) * COB;
  const int64_t o0 = (int64_t)tile * M;
  const int g0 = tile * (M / SCN_TILE_GRANULE);
  const int g1 = min(g0 + M / SCN_TILE_GRANULE, n_gran);
  // this piece's region of the global pass table (SK: two per workgroup, alternating: wave 0 may write the next piece's table
  // while wave 1 still walks this one; the set-up barrier keeps it from getting two ahead)
  unsigned long long *const gt_region =
      GT ? gtab + (SK ? (size_t)(blockIdx.x * 2 + gpar) : (size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * (size_t)gstride : nullptr;
  gpar ^= 1;
  // output-tile BYTE address of (row dst, this lane's first plane) = dst * (OS * 4) + tile0, modulo 2^32: dst - o0 is what
  // counts (kept in a VGPR: a VOP3 instruction takes one scalar operand)
  typedef __attribute__((address_space(3))) floatx4 lds_floatx4;
  unsigned tile0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)o_tile + (unsigned)(32 * w + 4 * lq) * 4u -
                   (unsigned)o0 * (unsigned)(OS * 4);
  asm volatile("" : "+v"(tile0));
  const unsigned li8[2] = {(unsigned)li * 8u, (unsigned)(li + 16) * 8u};  // byte offset of the lane's pair within a pass

  // each wave initialises ITS 32 planes of the output tile.  With a residual the 16 row pieces of a lane are requested
  // TOGETHER (the first version waited for every piece before asking for the next: sixteen HBM round trips per tile, 10-20 %
  // of the launches that carry a shortcut -- bench.py's `layers` showed forward launches 10 % behind the dInput launches
  // of the same shape)
  {
    constexpr int NIT = M * 8 / 64;  // row pieces per lane; i = lane + 64 j: the plane quad (i & 7) is the lane's own
    const int c4 = (lane & 7) * 4;
    floatx4 bv = {0.f, 0.f, 0.f, 0.f}, rv[NIT];
    if (SK && seed_slab >= 0) {
      // the item's first passes were computed by the workgroup that holds the previous slice: start from its tile.  That
      // piece ran FIRST in its slice and this one runs LAST in ours, so the flag is up long before it is looked at.  The slab
      // was stored write-through (sc1) and is read with sc1 loads: the L2 caches of two XCDs are not coherent.
      if (tid == 0) {
        unsigned spins = 0;  // (bounded, ~1 s: a launch that can never finish takes the whole GPU down; word 9 says it gave up)
        while (__hip_atomic_load(sk_flags + seed_slab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1u << 23))
          __builtin_amdgcn_s_sleep(4);
        if (spins >= (1u << 23)) __hip_atomic_store(sk_state + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SCN_TUNING
        atomicAdd(&g_stamps[6], (unsigned long long)spins);  // (tools/ab_stream_k.py --spins: how long seeded pieces wait)
        atomicAdd(&g_stamps[7], 1ull);
#endif
        __hip_atomic_store(sk_flags + seed_slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (one reader per flag)
      }
      __syncthreads();
      asm volatile("" ::: "memory");
      const __amdgpu_buffer_rsrc_t sl_rsrc = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(slabs + (size_t)seed_slab * (size_t)(M * COB)), 0, M * COB * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(sl_rsrc, ((w * NIT + j) * 64 + lane) * 16, 0, 16 /* sc1 */);
        rv[j] = floatx4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    } else {
    if (bias && blockIdx.y == 0) bv = *(const floatx4 *)(bias + co0 + 32 * w + c4);
    if (residual && blockIdx.y == 0) {
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        // (rows past the end of the tensor repeat its last row: they are never stored, and sixteen loads behind sixteen
        //  branches are 2^16 paths for tools/check_asm_loads.py)
        const int64_t r = min(o0 + (lane >> 3) + 8 * j, n_out - 1);
        rv[j] = *(const floatx4 *)(residual + r * Co + co0 + 32 * w + c4);
      }
      // (SK: hipcc lays this arm out in front of the seeded one and skips that through a scalar flag; the checker, which
      //  follows every branch, would see these loads in flight under the slab loads.  The tile writes below need them anyway.)
      if (SK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int j = 0; j < NIT; ++j) rv[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) *(floatx4 *)(o_tile + ((lane >> 3) + 8 * j) * OS + 32 * w + c4) = rv[j] + bv;
    if (!SK && lane < 8) *(floatx4 *)(o_tile + M * OS + 32 * w + c4) = floatx4{0.f, 0.f, 0.f, 0.f};  // (row M: unused sink of older builds)
  }
  if (w == 0) {  // pass table: lane k of wave 0 tabulates the passes of offset k at the position given by a wave scan.  (`w` is a
    // scalar: a branch, not an EXEC mask -- hipcc placed register copies and constants that wave 1 needs later in front of the
    // `s_or_b64 exec` that ended the `tid < 64` form of this region, see the note at the assembly pass loop)
    int s = 0, c = 0;
    // ksplit > 1 (launches with few tiles, the coarse UNet levels): this workgroup takes the offsets k == blockIdx.y
    // (mod ksplit) and writes its partial tile to slab blockIdx.y of the launcher's workspace (k_split_reduce adds them)
    if (tid < K && tid % ksplit == (int)blockIdx.y) {
      s = bases[(int64_t)tid * (n_gran + 1) + g0];
      c = bases[(int64_t)tid * (n_gran + 1) + g1] - s;
    }
    const int npl = (c + 31) >> 5;  // passes per chunk
    int pos = npl * nch;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = ASM ? shfl_i(pos, lane - o) : __shfl_up(pos, o);
      if (lane >= o) pos += y;
    }
    const int total_item = ASM ? shfl_i(pos, 63) : __shfl(pos, 63);
    pos -= npl * nch;
    // (SK: the piece holds the entries [p_lo, p_hi) of the item's table, stored from index 0)
    const int t_lo = SK ? min(p_lo, total_item) : 0, t_hi = SK ? min(p_hi, total_item) : total_item;
    const int total = t_hi - t_lo;
    const unsigned first = (unsigned)(((int64_t)tid * stride + s) * 8);
    uint2 *gt = (uint2 *)gt_region;
    for (int chn = 0; chn < nch; ++chn)
      for (int j = 0; j < npl; ++j) {
        const unsigned e0 = first + (unsigned)j * 256u;
        const unsigned e1 = (unsigned)(c - 32 * j) | (unsigned)chn << 8 | (unsigned)tid << 12 | (j == 0 ? 1u << 20 : 0u);
        if (ASM) {
          // 16-byte entries: + the byte offset of the phase's first weight block (wave 0's) and the lane masks of the pass' two groups
          if (!SK || (pos >= t_lo && pos < t_hi)) {
            const int left = c - 32 * j;
            const unsigned m0 = left >= 16 ? 0xffffu : (1u << left) - 1u;
            const unsigned m1 = left <= 16 ? 0u : left >= 32 ? 0xffffu : (1u << (left - 16)) - 1u;
            const unsigned woff = (unsigned)(((tid * (Ci / 16) + chn * (CK / 16)) * (Co / 16) + co0 / 16) * 1024);
            ((uint4 *)gt_region)[pos - t_lo] = make_uint4(e0, e1, woff, m0 | m1 << 16);
          }
        } else if (GT) { if (!SK || (pos >= t_lo && pos < t_hi)) gt[pos - t_lo] = make_uint2(e0, e1); }
        else { ptab[pos][0] = e0; ptab[pos][1] = e1; }
        ++pos;
      }
    if (tid < 8) {  // sentinels: empty passes
      if (ASM) {
        // (a zero that hipcc cannot hoist out of the piece loop: as a loop-invariant vector it was kept in four registers across
        //  the assembly block -- spilled to scratch)
        const unsigned zz = (unsigned)tid >> 31;
        ((uint4 *)gt_region)[total + tid] = make_uint4(zz, zz, zz, zz);
      }
      else if (GT) gt[total + tid] = make_uint2(0u, 0u);
      else { ptab[total + tid][0] = 0u; ptab[total + tid][1] = 0u; }
    }
    if (GT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the table is in L2 before the barrier lets anybody read it
  }
  __syncthreads();  // the only barrier
  // (the scalar cache is not coherent with this kernel's vector stores; the invalidate is a scalar-memory operation like the
  //  loads behind it and those may complete out of order: wait for it)
  if (GT) asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long gt_s = 0;  // this workgroup's region, in a scalar register pair
  if (GT) {
    const unsigned long long a = (unsigned long long)gt_region;
    gt_s = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32;
  }
  // offset of the loads a pass does not need (absent second group, passes past the end): the first bytes of the buffer
  // (always valid; what they fetch is multiplied by nothing or lands in the sink row).  Pointing them past the end of the
  // buffer instead (DBG & 8192: the range check returns 0 without touching memory) is 0.5-2 % SLOWER.
  constexpr unsigned OOB = (DBG & 8192) ? 0x80000000u : 0u;
  struct Ent { unsigned base, meta; };  // meta & 255: pairs left (0: no such pass)
  auto entry = [&](int p) __attribute__((always_inline)) {
    if (GT) {
      unsigned long long v;
      const int off = uni(p) * 8;
      asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(gt_s), "s"(off) : "memory");
      return Ent{(unsigned)v, (unsigned)(v >> 32)};
    }
    const uint2 v = *(const uint2 *)ptab[p];
    return Ent{(unsigned)uni((int)v.x), (unsigned)uni((int)v.y)};
  };
  auto rem = [](const Ent &e) __attribute__((always_inline)) { return (int)(e.meta & 255u); };
  // per-lane byte offsets of a pass's two index loads
  auto index_offsets = [&](const Ent &e, unsigned (&vo)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // the lane's pair, clamped to the last pair of the list: lanes past the end (a whole absent
      // second group too) repeat it -- rows that are in the cache anyway -- and are masked out of the output-tile write;
      // rem == 0 (no such pass): no clamp, the first pairs of `rules`
      if (DBG & 4096) vo[i] = OOB;
      else vo[i] = min(li8[i], (unsigned)(rem(e) - 1) * 8u);
    }
  };
  // row byte offsets / output-tile rows of a pass from its pair indices
  auto addr = [&](const Ent &e, const u32x2 (&pr)[2], unsigned (&off)[2], int (&orow)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // one v_mad_u32_u24 each (rows and sites are < 2^24, conv_plan): beside f32 MFMAs every vector-ALU instruction costs
      // 8-12 cycles of matrix-pipe time (tools/mfma_probe.py), the quarter-rate 32-bit multiplies no less
      const unsigned src = icol ? pr[i].y : pr[i].x;
      const unsigned dst = icol ? pr[i].x : pr[i].y;
      off[i] = (DBG & 2) ? OOB : mad24(src, row_bytes, lq16);
      orow[i] = (int)mad24(dst, os_u, tile0);
    }
  };
  auto weight_scalars = [&](const Ent &e, int (&so)[NS16], unsigned &voff, bool &fresh) __attribute__((always_inline)) {
    const int k = (int)(e.meta >> 12) & 255, chn = (int)(e.meta >> 8) & 15;
    fresh = (e.meta >> 20) & 1u;
    const int blk = (k * (Ci / 16) + chn * (CK / 16)) * (Co / 16) + co0 / 16 + w * COS;
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp) so[sp] = (blk + sp * (Co / 16)) * 1024;
    voff = (DBG & 8) ? OOB : (unsigned)lane * 16u;  // (the loads themselves are conditional)
  };
  auto load_idx = [&](const Ent &e, const unsigned (&vo)[2], u32x2 (&pr)[2]) __attribute__((always_inline)) {
    const int base = (int)e.base;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(pr[i]) : "v"(vo[i]), "s"(ru_rsrc), "s"(base) : "memory");
  };
  auto load_rows = [&](int i, unsigned voff, int so, floatx4 (&rows)[2][NS16]) __attribute__((always_inline)) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rows[i][0]) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:64" : "=v"(rows[i][1]) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:128" : "=v"(rows[i][2]) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:192" : "=v"(rows[i][3]) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
  };
  auto load_weights = [&](int sp, unsigned voff, int so, floatx4 (&wr)[COS][NS16]) __attribute__((always_inline)) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(wr[0][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(wr[1][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
  };
#define SCN_ROWS(R) "v"(R[0][0]), "v"(R[0][1]), "v"(R[0][2]), "v"(R[0][3]), "v"(R[1][0]), "v"(R[1][1]), "v"(R[1][2]), "v"(R[1][3])
  auto scatter = [&](const floatx4 (&a)[2][COS], const floatx4 (&old)[2][COS], const int (&orow)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < COS; ++q) {
        if (DBG & 1) {
          asm volatile("" ::"v"(a[i][q]));
        } else {  // absent group / padded pair: sink row, += 0 or garbage nobody reads
          *(floatx4 *)__builtin_assume_aligned(o_tile + orow[i] + 32 * w + 16 * q + 4 * lq, 16) = old[i][q] + a[i][q];
        }
      }
  };
  auto tile_rows = [&](floatx4 (&old)[2][COS], const int (&orow)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < COS; ++q)
        // (without the assumption hipcc loses the 16-byte alignment through the rotating row variables and emits two
        //  ds_read2_b32 / ds_write2_b32 per fragment instead of one ds_read_b128 / ds_write_b128)
        if (!(DBG & 1)) old[i][q] = *(const floatx4 *)__builtin_assume_aligned(o_tile + orow[i] + 32 * w + 16 * q + 4 * lq, 16);
  };
  // EXEC mask of the lanes whose pair exists: group g of a pass with r pairs left holds min(max(r - 16 g, 0), 16) pairs, lanes
  // li + 16 lq: the same 16 bits in all four quarters.  Scalar instructions only -- the sink-row redirect this replaces was
  // a v_cmp + v_cndmask per group.
  auto lane_mask = [](int r, int g) __attribute__((always_inline)) {
    // (written with compares and selects: hipcc turns max(r - 16, 0) into a saturating v_sub, a vector instruction)
    const int n = r <= 16 * g ? 0 : (r >= 16 * g + 16 ? 16 : r - 16 * g);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(((1u << n) - 1u) * 0x10001u));
  };
  // a + b as two v_pk_add_f32 (hipcc splits about half of these vector adds into four v_add_f32)
  auto add4 = [](const floatx4 &x, const floatx4 &y) __attribute__((always_inline)) {
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    floatx2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(floatx2{x.x, x.y}), "v"(floatx2{y.x, y.y}));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(floatx2{x.z, x.w}), "v"(floatx2{y.z, y.w}));
    return floatx4{lo.x, lo.y, hi.x, hi.y};
  };
  auto tile_read2 = [&](int addr, floatx4 (&o)[COS]) __attribute__((always_inline)) {
    const lds_floatx4 *p = (const lds_floatx4 *)(size_t)(unsigned)addr;
    o[0] = p[0];
    o[1] = p[4];
  };
  auto tile_write2 = [&](int addr, unsigned mask, const floatx4 &d0, const floatx4 &d1) __attribute__((always_inline)) {
    asm volatile(
        "s_mov_b32 exec_lo, %[m]\n\t"
        "s_mov_b32 exec_hi, %[m]\n\t"
        "ds_write_b128 %[a], %[d0]\n\t"
        "ds_write_b128 %[a], %[d1] offset:64\n\t"
        "s_mov_b64 exec, -1"
        :
        : [a] "v"(addr), [m] "s"(mask), [d0] "v"(d0), [d1] "v"(d1)
        : "memory");
  };

  // SK, whole items: the NEXT item is drawn here -- the returning atomic travels with this piece's first index loads and the
  // wait behind them covers it (drawn at the top of the next piece it cost every item a round trip through a memory path
  // that a CU's eight gathering waves keep full: 2-3 us per dependent access)
  // (sk_phase == 0: the next call of sk_next begins with a draw.  Only draws of WHOLE items are made ahead: a slice drawn while
  //  its workgroup still has a piece to finish sits undone, and the slice behind it waits for its first piece -- measured:
  //  launches 7-12 % slower)
  const bool pf_now = SK && SK_PREFETCH_DRAW && !sk_static && sk_phase == 0 && !sk_steal_a && na_items > 0 && na_items_a != (int)0x80000000;
  const int pf_slot = sk_par, pf_which = 1;
  unsigned pf_val = 0;
  if (pf_now && tid == 0)
    pf_val = __hip_atomic_fetch_add(sk_state + (pf_which == 1 ? 0 : 16) + ((int)blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (SK) { sk_pf = pf_now ? pf_which : 0; tile_parked = tile; park_all(true); }
  if (ASM) {
    // the whole pass loop, prologue and last pending write included (tile4_loop_asm.inc).
    // NO per-lane value may be live across this block or reach it from the set-up above.  The block clobbers v60..v255, so hipcc
    // moves whatever vector value it needs later into a low register beforehand -- and it placed such a copy at the top of the
    // block that joins the `tid < 64` table build, IN FRONT of the `s_or_b64 exec` that re-enables wave 1 (round 6, found with
    // rocgdb: wave 1 of a stream-K workgroup kept the previous piece's register contents as its "(lane & 7) * 4" and loaded the
    // batch-norm statistics of the epilogue from a wild address).  Hence: the lane id is taken from the hardware right here
    // (v_mbcnt, volatile asm: nothing derived from it can be computed earlier) for the loop's operands, and again behind the
    // block for everything the epilogue needs; scalars cross in scalar registers / v_writelane lanes, which do not depend on EXEC.
    int l2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
    const unsigned wstep = (unsigned)(Co / 16) * 1024u, wbase = (unsigned)(w * COS) * 1024u;
    const unsigned li_2 = (unsigned)l2 & 15u, lq_2 = (unsigned)l2 >> 4;
    unsigned wlane16 = (unsigned)l2 * 16u, lq16_v = lq_2 * 16u, li8_0 = li_2 * 8u, li8_1 = (li_2 + 16u) * 8u;
    unsigned tile0_2 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)o_tile + ((unsigned)(32 * w) + 4u * lq_2) * 4u -
                       (unsigned)o0 * (unsigned)(OS * 4);
#define SCN_T4_ASM_OPERANDS                                                                                                         \
  :                                                                                                                                 \
  : [in] "s"(in_rsrc), [wp] "s"(wp_rsrc), [ru] "s"(ru_rsrc), [gt] "s"(gt_s), [row_bytes] "s"(row_bytes), [os_u] "s"(os_u),         \
    [wstep] "s"(wstep), [wbase] "s"(wbase), [tile0] "v"(tile0_2), [wlane16] "v"(wlane16), [lq16] "v"(lq16_v), [li8_0] "v"(li8_0),  \
    [li8_1] "v"(li8_1)                                                                                                              \
  : SCN_T4_ASM_CLOBBERS
    if (ASM == 1 && INCOL == 0) asm volatile(SCN_T4_ASM_INCOL0 SCN_T4_ASM_OPERANDS);
    else if (ASM == 1) asm volatile(SCN_T4_ASM_INCOL1 SCN_T4_ASM_OPERANDS);
#ifdef SCN_TUNING  // timing-only ablations of the loop (tools/gen_tile4_loop.py ABLATIONS; tools/bench_conv.py --asm)
    else if (ASM == 2) asm volatile(SCN_T4_ASM_ABL2 SCN_T4_ASM_OPERANDS);
    else if (ASM == 3) asm volatile(SCN_T4_ASM_ABL3 SCN_T4_ASM_OPERANDS);
    else if (ASM == 4) asm volatile(SCN_T4_ASM_ABL4 SCN_T4_ASM_OPERANDS);
    else if (ASM == 5) asm volatile(SCN_T4_ASM_ABL5 SCN_T4_ASM_OPERANDS);
    else if (ASM == 6) asm volatile(SCN_T4_ASM_ABL6 SCN_T4_ASM_OPERANDS);
    else if (ASM == 9) asm volatile(SCN_T4_ASM_ABL9 SCN_T4_ASM_OPERANDS);
#endif
#undef SCN_T4_ASM_OPERANDS
    int l3;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l3));
    lane = l3;
    tid = 64 * w + l3;
  }
  Ent E0 = ASM ? Ent{0u, 0u} : entry(0);
  if (SK && pf_now && !(rem(E0) > 0) && tid == 0) sk_slot[pf_slot] = (int)pf_val;  // (an item without passes: no prologue)
  if (rem(E0) > 0) {
    // ---- pipeline state at the top of pass p: rows(p) in rc, rows(p+1) in rm (arrived), weights(p) in wc, indices of
    // pass p+2 in pr (arrived); E0..E3 = table entries p..p+3; `ioff` = index-load offsets of pass p+3; acc[1] / op1 =
    // result and output-tile rows of the second group of pass p-1, not yet in the tile (the accumulators are never copied:
    // group 1 goes to the tile beside group 0 of the next pass, group 0 beside group 1 of its own).
    u32x2 pr[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};
    unsigned off[2], ioff[2];
    floatx4 rA[2][NS16], rB[2][NS16], rC[2][NS16], wc[COS][NS16], wn[COS][NS16], acc[2][COS];
    int oA[2], oB[2], oC[2], op1 = (int)tile0;  // output-tile byte addresses of passes p, p+1, p+2 (rotating with the row sets);
                                                // of group 1 of pass p-1
    unsigned mp1 = 0u;                          // its lane mask
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < COS; ++q) acc[i][q] = floatx4{0.f, 0.f, 0.f, 0.f};
    Ent E1 = entry(1), E2 = entry(2), E3 = entry(3);
    int pnext = 4;  // table index of the entry read during the current pass
    {  // prologue (not overlapped; once per tile)
      int so[NS16];
      unsigned voff;
      bool fresh;
      index_offsets(E0, ioff);
      load_idx(E0, ioff, pr);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(pr[0]), "v"(pr[1]) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (SK && pf_now) {  // (the draw made above has returned)
        if (tid == 0) sk_slot[pf_slot] = (int)pf_val;
        __builtin_amdgcn_sched_barrier(0);
      }
      addr(E0, pr, off, oA);
      weight_scalars(E0, so, voff, fresh);
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp) load_weights(sp, (DBG & 8) ? OOB : (unsigned)lane * 16u, so[sp], wc);
      load_rows(0, off[0], (int)((E0.meta >> 8) & 15u) * (CK * 4), rA);
      load_rows(1, off[1], (int)((E0.meta >> 8) & 15u) * (CK * 4), rA);
      __builtin_amdgcn_sched_barrier(0);
      index_offsets(E1, ioff);
      load_idx(E1, ioff, pr);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(pr[0]), "v"(pr[1]), SCN_ROWS(rA), SCN_ROWS(wc) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      addr(E1, pr, off, oB);
      load_rows(0, off[0], (int)((E1.meta >> 8) & 15u) * (CK * 4), rB);
      load_rows(1, off[1], (int)((E1.meta >> 8) & 15u) * (CK * 4), rB);
      __builtin_amdgcn_sched_barrier(0);
      index_offsets(E2, ioff);
      load_idx(E2, ioff, pr);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(pr[0]), "v"(pr[1]), SCN_ROWS(rB) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      index_offsets(E3, ioff);
    }

    unsigned wlane16 = (DBG & 8) ? OOB : (unsigned)lane * 16u;
    asm volatile("" : "+v"(wlane16));  // (one register for the whole loop, not rematerialised)
    auto body = [&](floatx4 (&rc)[2][NS16], floatx4 (&rm)[2][NS16], floatx4 (&rn)[2][NS16], int (&oc)[2], int (&on)[2]) __attribute__((always_inline)) {
      const bool two = rem(E0) > 16;
      const unsigned m0 = lane_mask(rem(E0), 0), m1 = lane_mask(rem(E0), 1);
      floatx4 old[2][COS];
      Ent E4 = E3;
      unsigned long long eraw = 0;
      int w_so[NS16], rso = 0;
      unsigned w_off_unused = OOB;
      bool w_new = false;
      // (the accumulators are not zeroed: the first MFMA of a chain takes C = 0; those of an absent second group keep what
      //  they held, which goes to the sink row)
      // Everything that is not an MFMA of this pass, cut into pieces of a few instructions; piece m is issued right
      // after MFMA m of the first group (16 pieces).  Measured the same as seven larger slots between groups of four
      // MFMAs (MFMA pipe busy 66 % either way, two waves per SIMD): the granularity of the interleave is not what
      // limits the kernel, the vector-memory issue of the 10-18 loads per pass is.  Load order (vmcnt!): 8 weight,
      // 2 index, 8 row.
      // (the "next pass starts a phase" bit, re-read through an opaque scalar move at every use: a common subexpression
      //  across the pieces ends up as a 0/1 VGPR, v_cndmask + v_cmp per pass)
      auto next_is_new = [&]() __attribute__((always_inline)) {
        unsigned m_;
        asm volatile("s_mov_b32 %0, %1" : "=s"(m_) : "s"(E1.meta));
        return ((m_ >> 20) & 1u) != 0u;
      };
      auto piece = [&](int m) __attribute__((always_inline)) {
        if (m == 0 || m == 1) {  // row offset / output row of group m of pass p+2 from its (arrived) pair indices
          const int i = m;
          const unsigned src = icol ? pr[i].y : pr[i].x;
          const unsigned dst = icol ? pr[i].x : pr[i].y;
          off[i] = ((DBG & 2) || ((DBG & 2048) && w == 1)) ? OOB : mad24(src, row_bytes, lq16);
          on[i] = (int)mad24(dst, os_u, tile0);
          if (m == 1) rso = (int)((E2.meta >> 8) & 15u) * (CK * 4);
        } else if (m == 2) {     // weights of pass p+1 only when it starts a new (offset, chunk) phase: they are OLDER than
          // the row loads, so the vmcnt(8) below is right either way.  (The condition is re-read from the scalar table entry
          // every time: kept in a bool across the pieces, hipcc parks it in a VGPR -- v_cndmask + v_cmp per pass.)
          if (next_is_new()) weight_scalars(E1, w_so, w_off_unused, w_new);
        } else if (m >= 3 && m <= 6) {
          // (the per-lane offset is loop invariant: as a variable set inside the branch above it cost a v_mov per pass)
          if (next_is_new()) load_weights(m - 3, wlane16, w_so[m - 3], wn);
        } else if (m == 7) {     // group 1 of the previous pass -> output tile: read its rows ...
          if (!(DBG & 1)) tile_read2(op1, old[1]);
        } else if (m == 8) {     // indices of pass p+3 (pr was consumed by pieces 0-1)
          load_idx(E3, ioff, pr);
          if (GT) {  // table entry p+4 requested here, picked up in piece 15 (a scalar load: nothing else names `eraw` meanwhile)
            const int eoff = uni(pnext) * 8;
            asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(eraw) : "s"(gt_s), "s"(eoff) : "memory");
          }
        } else if (m >= 9 && m <= 12) {  // rows of pass p+2: the long-latency gathers, in flight until the end of pass p+1
          const int i = (m - 9) >> 1;
          if ((m - 9) & 1) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:128" : "=v"(rn[i][2]) : "v"(off[i]), "s"(in_rsrc), "s"(rso) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:192" : "=v"(rn[i][3]) : "v"(off[i]), "s"(in_rsrc), "s"(rso) : "memory");
          } else {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rn[i][0]) : "v"(off[i]), "s"(in_rsrc), "s"(rso) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:64" : "=v"(rn[i][1]) : "v"(off[i]), "s"(in_rsrc), "s"(rso) : "memory");
          }
        } else if (m == 13) {    // ... add, write (its accumulators are free again before this pass's second group starts)
          if (DBG & 1) asm volatile("" ::"v"(acc[1][0]), "v"(acc[1][1]));
          else tile_write2(op1, mp1, add4(old[1][0], acc[1][0]), add4(old[1][1], acc[1][1]));
        } else if (m == 14) {    // group 0 of THIS pass: read its rows (added and written beside the second group's MFMAs)
          if (!(DBG & 1)) tile_read2(oc[0], old[0]);
        } else if (m == 15) {    // table entry p+4
          if (GT) {
            unsigned long long ev;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b64 %0, %1" : "=s"(ev) : "s"(eraw) : "memory");
            E4 = Ent{(unsigned)ev, (unsigned)(ev >> 32)};
          } else {
            E4 = entry(pnext);
          }
        }
      };
      if (!(DBG & 16)) {
#pragma unroll
        for (int sp = 0; sp < NS16; ++sp) {
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            const int q = h & 1;
            const float a = (h >> 1) == 0 ? wc[q][sp].x : (h >> 1) == 1 ? wc[q][sp].y : (h >> 1) == 2 ? wc[q][sp].z : wc[q][sp].w;
            const float bv = (h >> 1) == 0 ? rc[0][sp].x : (h >> 1) == 1 ? rc[0][sp].y : (h >> 1) == 2 ? rc[0][sp].z : rc[0][sp].w;
            acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, (sp == 0 && h < 2) ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[0][q], 0, 0, 0);
            if (sp * 8 + h < 16) {
              __builtin_amdgcn_sched_barrier(0);
              if (DBG & 256) t0 = stamp();
              piece(sp * 8 + h);
              if (DBG & 256) { t1 = stamp(); seg[(sp * 8 + h) < 3 ? 0 : (sp * 8 + h) < 8 ? 1 : (sp * 8 + h) < 9 ? 2 : 3] += t1 - t0; }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) piece(m);
      }
      auto write_group0 = [&]() __attribute__((always_inline)) {
        if (DBG & 1) asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][1]));
        else tile_write2(oc[0], m0, add4(old[0][0], acc[0][0]), add4(old[0][1], acc[0][1]));
      };
      if (!two || (DBG & 16)) write_group0();
      if (two && !(DBG & 16)) {
#pragma unroll
        for (int sp = 0; sp < NS16; ++sp) {
          if (sp == 1) {  // eight MFMAs behind the end of group 0: its result has landed
            __builtin_amdgcn_sched_barrier(0);
            write_group0();
            __builtin_amdgcn_sched_barrier(0);
          }
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0][sp].x, rc[1][sp].x, sp == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[1][sp].x, rc[1][sp].x, sp == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[1][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0][sp].y, rc[1][sp].y, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[1][sp].y, rc[1][sp].y, acc[1][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0][sp].z, rc[1][sp].z, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[1][sp].z, rc[1][sp].z, acc[1][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0][sp].w, rc[1][sp].w, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[1][sp].w, rc[1][sp].w, acc[1][1], 0, 0, 0);
        }
      }
      // rows of pass p+1 (requested one pass ago: rm), weights of pass p+1 and indices of pass p+3 (this pass); the 8 row
      // loads of pass p+2 stay in flight.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(8)" ::SCN_ROWS(rm), SCN_ROWS(wn), "v"(pr[0]), "v"(pr[1]) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (DBG & 256) seg[4] += 1;
      op1 = oc[1];
      mp1 = m1;
      if (next_is_new()) {
#pragma unroll
        for (int q = 0; q < COS; ++q)
#pragma unroll
          for (int sp = 0; sp < NS16; ++sp) wc[q][sp] = wn[q][sp];
      }
      E0 = E1; E1 = E2; E2 = E3; E3 = E4;
      index_offsets(E3, ioff);
      ++pnext;
    };
    do {  // single exit: a mid-loop break makes hipcc read "undefined" scalars out of a row register.  A pass past the
          // end is an empty one (dummy loads, output rows = sink) that only flushes `prev`.
      body(rA, rB, rC, oA, oC);
      body(rB, rC, rA, oB, oA);
      body(rC, rA, rB, oC, oB);
    } while (rem(E0) > 0);
    // the (dummy) row prefetches of the passes past the end
    asm volatile("s_waitcnt vmcnt(0)" ::SCN_ROWS(rA), SCN_ROWS(rB), SCN_ROWS(rC) : "memory");
    if (!(DBG & 1)) {  // group 1 of the last pass
      floatx4 o1[COS];
      tile_read2(op1, o1);
      tile_write2(op1, mp1, add4(o1[0], acc[1][0]), add4(o1[1], acc[1][1]));
    }
  }
#undef SCN_ROWS
  if (SK) park_all(false);
  const int tile_e = SK ? tile_parked : tile;  // (SK: `tile` and `o0_e` are not kept across the pass loop)
  const int64_t o0_e = (int64_t)tile_e * M;
  if (SK && out_slab >= 0) {
    // the item goes on in the next slice: publish the tile (each wave its 32 planes, in the order the seeded set-up reads
    // them back: piece j of lane l = row (l >> 3) + 8 j, plane quad l & 7).  Write-through stores, every storing wave drains
    // them, the barrier, then ONE lane raises the flag (sc1 store): the hand-off form the other XCDs' L2s require.
    const __amdgpu_buffer_rsrc_t sl_rsrc = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(slabs + (size_t)out_slab * (size_t)(M * COB)), 0, M * COB * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < M * 8 / 64; ++j) {
      const floatx4 v = *(const floatx4 *)(o_tile + ((lane >> 3) + 8 * j) * OS + 32 * w + (lane & 7) * 4);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)},
                                             sl_rsrc, ((w * (M * 8 / 64) + j) * 64 + lane) * 16, 0, 16 /* sc1 */);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(sk_flags + out_slab, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    continue;
  }
  // each wave stores its own 32 planes: no barrier needed
  const int rows = (int)min((int64_t)M, n_out - o0_e);
  float *dst = out + (int64_t)blockIdx.y * n_out * Co;
  if (SPLIT && ksplit > 1) {
    // ---- split launch, partial tiles added by the tile's last workgroup (see SPLIT above).  The hand-over that the L2 caches
    // of eight XCDs require: write-through (sc1) stores of the partial tile, every wave drains them, the barrier, ONE lane draws
    // the item's ticket; the workgroup that draws the last one reads all slabs with sc1 loads.  The ticket word goes back to 0.
    const int co_w = (item % ncb) * COB + 32 * w;
    const size_t slab_elems = (size_t)n_out * (size_t)Co;
    const unsigned row_b = (unsigned)Co * 4u;
    const unsigned tile_b = (unsigned)(((size_t)o0_e * (size_t)Co + (size_t)co_w) * 4);  // (< 2 GiB: the launcher checks)
    {
      const __amdgpu_buffer_rsrc_t ds = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(dst), 0, (int)(slab_elems * 4), 0x00020000);
      for (int i = lane; i < rows * 8; i += 64) {
        const int r = i >> 3, c4 = (i & 7) * 4;
        const floatx4 v = *(const floatx4 *)(o_tile + r * OS + 32 * w + c4);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, ds,
                                               tile_b + (unsigned)r * row_b + (unsigned)c4 * 4u, 0, 16 /* sc1 */);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *const tk_slot = (int *)(o_tile + M * OS);  // (row M: the unused sink row)
    if (tid == 0) tk_slot[0] = (int)__hip_atomic_fetch_add(sk_state + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (uni(tk_slot[0]) != ksplit - 1) return;
    if (tid == 0) __hip_atomic_store(sk_state + item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");
    // slab 0 first -- it holds bias and shortcut --, then 1, 2, ...: k_split_reduce's order.  Three slabs' worth of loads in
    // flight, eight 16-byte pieces per lane each; a wave adds, stores and sums ITS 32 planes (no barrier from here on).
    float *const fin = slabs;
    const bool st_on = colstats.out != nullptr;
    floatx4 cs = {0.f, 0.f, 0.f, 0.f}, css = {0.f, 0.f, 0.f, 0.f};
    StatsCols sc{};  // (value-initialised: left undefined, its twelve registers became state carried from piece to piece -- spilled)
    sc.bn = false;
    if (st_on) sc.init(colstats, co_w + (lane & 7) * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      constexpr int NE = 8;
      floatx4 sum[NE];
      unsigned off[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int i = lane + 64 * (h * NE + e), r = i >> 3, c4 = (i & 7) * 4;
        off[e] = r < rows ? tile_b + (unsigned)r * row_b + (unsigned)c4 * 4u : 0x80000000u;  // (out of range: reads zeros)
      }
      // (every load that is issued is consumed: a load left in flight over the loop's back edge is what tools/check_asm_loads.py
      //  exists to find.  ksplit is 3, 9 or 27, or 8 for a strided rulebook: whole groups of three, then single slabs)
      auto slab_rsrc = [&](int j) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(uni_ptr(out + (size_t)j * slab_elems), 0, (int)(slab_elems * 4), 0x00020000);
      };
      auto as_f4 = [](const u32x4 &v) __attribute__((always_inline)) {
        return floatx4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      };
      int j0 = 0;
      for (; j0 + 3 <= ksplit; j0 += 3) {
        u32x4 b[3][NE];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const __amdgpu_buffer_rsrc_t ps = slab_rsrc(j0 + u);
#pragma unroll
          for (int e = 0; e < NE; ++e) b[u][e] = __builtin_amdgcn_raw_buffer_load_b128(ps, off[e], 0, 16 /* sc1 */);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int e = 0; e < NE; ++e) sum[e] = (j0 + u == 0) ? as_f4(b[u][e]) : sum[e] + as_f4(b[u][e]);
      }
      for (; j0 < ksplit; ++j0) {
        u32x4 b[NE];
        const __amdgpu_buffer_rsrc_t ps = slab_rsrc(j0);
#pragma unroll
        for (int e = 0; e < NE; ++e) b[e] = __builtin_amdgcn_raw_buffer_load_b128(ps, off[e], 0, 16 /* sc1 */);
#pragma unroll
        for (int e = 0; e < NE; ++e) sum[e] = (j0 == 0) ? as_f4(b[e]) : sum[e] + as_f4(b[e]);
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int i = lane + 64 * (h * NE + e), r = i >> 3, c4 = (i & 7) * 4;
        if (r < rows) {
          *(floatx4 *)(fin + (o0_e + r) * Co + co_w + c4) = sum[e];
          if (st_on) sc.add(sum[e], colstats.x + (o0_e + r) * Co + co_w + c4, cs, css);
        }
      }
    }
    if (st_on) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cs[e] += ASM ? shfl_f(cs[e], lane ^ o) : __shfl_xor(cs[e], o);
          css[e] += ASM ? shfl_f(css[e], lane ^ o) : __shfl_xor(css[e], o);
        }
      if (lane < 8) {
        *(floatx4 *)(colstats.out + ((int64_t)tile_e * 2 + 0) * Co + co_w + lane * 4) = cs;
        *(floatx4 *)(colstats.out + ((int64_t)tile_e * 2 + 1) * Co + co_w + lane * 4) = css;
      }
    }
    return;
  }
  // ... and the column statistics of these 32 planes over the tile's rows come from the SAME 16-byte reads: lane (row
  // octet lane >> 3, plane quad lane & 7) sums its 16 rows, a butterfly over the eight lanes of a plane quad finishes.  (The
  // first version walked the tile again, one float per lane and row, 64 dependent LDS reads per lane: 3-6 % of a forward
  // launch inside the train step.)
  const bool stats = colstats.out != nullptr && ksplit == 1;
  floatx4 cs = {0.f, 0.f, 0.f, 0.f}, css = {0.f, 0.f, 0.f, 0.f};
  StatsCols sc{};  // (value-initialised: left undefined, its twelve registers became state carried from piece to piece -- spilled)
  sc.bn = false;
  if (stats) sc.init(colstats, co0 + 32 * w + (lane & 7) * 4);  // (i & 7 == lane & 7: a lane keeps its plane quad)
  for (int i = lane; i < rows * 8; i += 64) {
    const int r = i >> 3, c4 = (i & 7) * 4;
    const floatx4 v = *(const floatx4 *)(o_tile + r * OS + 32 * w + c4);
    *(floatx4 *)(dst + (o0_e + r) * Co + co0 + 32 * w + c4) = v;
    sc.add(v, colstats.x + (o0_e + r) * Co + co0 + 32 * w + c4, cs, css);
  }
  if (stats) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cs[e] += ASM ? shfl_f(cs[e], lane ^ o) : __shfl_xor(cs[e], o);
        css[e] += ASM ? shfl_f(css[e], lane ^ o) : __shfl_xor(css[e], o);
      }
    if (lane < 8) {
      *(floatx4 *)(colstats.out + ((int64_t)tile_e * 2 + 0) * Co + co0 + 32 * w + lane * 4) = cs;
      *(floatx4 *)(colstats.out + ((int64_t)tile_e * 2 + 1) * Co + co0 + 32 * w + lane * 4) = css;
    }
  }
  if (!SK) break;
  // (the set-up that follows waits for its first loads, hence -- vmcnt counts in order -- for these stores, within a
  //  microsecond anyway; said here, tools/check_asm_loads.py sees an empty queue at the top of every piece)
  if (SK_DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }  // pieces
  if ((DBG & 256) && lane == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&g_stamps[i], seg[i]);
  if (SK && thread0() && na_items_a != (int)0x80000000 && !sk_static_a) {  // the last workgroup out leaves the counters as it found them (every flag was reset by its reader)
    if (__hip_atomic_fetch_add(sk_state + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
#pragma unroll
      for (int i = 0; i < 11; ++i)
        if (i != 9) __hip_atomic_store(sk_state + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 16; i < 24; ++i) __hip_atomic_store(sk_state + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------- forward, barrier-free, few vector-ALU instructions (v6) ---
// k_conv_tile4's division of labour (two waves per tile, 32 planes each, rows gathered straight into MFMA fragment
// registers, output tile in LDS, pass table), rebuilt around one measurement (tools/mfma_probe.py): beside
// v_mfma_f32_16x16x4_f32 every vector-ALU instruction -- v_mov, v_cndmask, v_mad, v_accvgpr_read, v_pk_add alike -- takes
// 8 cycles of matrix-pipe time when it sits in a cluster of its kind and 12-14 when it stands alone between two MFMAs,
// with one wave per SIMD or with four; scalar instructions, branches, LDS and memory instructions cost 0-3 %.  k_conv_tile4
// issues ~90 vector-ALU instructions per pass of 32-64 MFMAs: that, not latency, is where a third of its matrix pipe
// goes.  Here a pass issues ~20, in three clusters:
//   * ONE row register set, reloaded in place (the 16-byte piece R[g][sp] of pass p is consumed by eight consecutive
//     MFMAs and requested again for pass p+1 right after them); indices converted and reloaded in place too: no
//     rotating register sets, no copies, no unrolling;
//   * addresses are one v_mad_u32_u24 each (row byte offset, output-tile byte address); lanes past the end of a list are
//     not redirected to a sink row (v_cmp + v_cndmask) but masked out of the output-tile WRITE with an EXEC mask made
//     by scalar instructions; absent groups are skipped by scalar branches;
//   * the accumulators are not copied or zeroed (first MFMA of a chain takes C = 0) and go to the output tile while
//     the OTHER group's MFMAs run;
//   * two weight register sets that swap roles from phase to phase -- the pass loop exists twice, one copy per role --
//     instead of 16 v_mov_b64 per phase.
// vmcnt is an in-order FIFO and every pass issues the same 18 loads in the same order (R0[0], 8 weight, R0[1], 2 index,
// R0[2], R0[3], R1[0..3]; the weight loads with EXEC = 0 unless the next pass starts a phase), so every wait is a
// constant: tools/check_asm_loads.py proves on the emitted assembly that no instruction touches a register with a load
// in flight.
template <int M, bool INCOL, int DBG = 0>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3)))
    k_conv_tile6(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ Wp,
                 const float *__restrict__ bias, const float *__restrict__ residual, StatsArg colstats, const int2 *__restrict__ rules,
                 const int32_t *__restrict__ bases, int K, int64_t stride,
                 int64_t n_out, int n_gran, int Ci, int Co, int ntiles, int ncb,
                 unsigned in_bytes, unsigned wp_bytes, int ksplit) {
  constexpr int COB = 64, OS = COB + 4, CK = 64, NS16 = 4, COS = 2;
  constexpr int TMAX = KMAX * (M / 32) * SCN_TILE4_MAX_CHUNKS + 8;  // passes per tile: <= M / 32 per (offset, chunk); + sentinels
  // dynamic LDS (the launcher passes tile6_lds_bytes<M>()): with a static 39 KB array hipcc concludes that three waves per
  // SIMD are out of reach anyway and allocates 250 registers with the accumulators in AGPRs
  extern __shared__ __attribute__((aligned(16))) float lds6[];
  float *o_tile = lds6;                                          // [M * OS]
  unsigned (*ptab)[2] = (unsigned (*)[2])(lds6 + M * OS);        // [TMAX][2], as in k_conv_tile4
  typedef __attribute__((address_space(3))) floatx4 lds_floatx4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds6;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Wp, 0, (int)wp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ru_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)rules, 0, (int)(K * stride * 8), 0x00020000);

  int lin;
  {  // workgroup -> tile: chunks of consecutive tiles per XCD, dealt round robin (see k_conv_tile4)
    const int total = ntiles * ncb;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int CH = 64 * ncb;
    const int full = (total / (8 * CH)) * (8 * CH);
    if (bid < full) {
      lin = ((j / CH) * 8 + xcd) * CH + j % CH;
    } else {
      const int rest = total - full, b2 = bid - full, x2 = b2 & 7, j2 = b2 >> 3;
      const int q = rest >> 3, r = rest & 7;
      lin = full + (x2 < r ? x2 * (q + 1) : r * (q + 1) + (x2 - r) * q) + j2;
    }
  }
  const int tile = lin / ncb;
  const int co0 = (lin % ncb) * COB;
  const int64_t o0 = (int64_t)tile * M;
  const int g0 = tile * (M / SCN_TILE_GRANULE);
  const int g1 = min(g0 + M / SCN_TILE_GRANULE, n_gran);
  const int nch = Ci / CK;

  for (int i = lane; i < M * 8; i += 64) {  // each wave initialises ITS 32 planes of the output tile
    const int r = i >> 3, c4 = (i & 7) * 4;
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
    if (bias && blockIdx.y == 0) v = *(const floatx4 *)(bias + co0 + 32 * w + c4);
    if (residual && blockIdx.y == 0 && o0 + r < n_out) v += *(const floatx4 *)(residual + (o0 + r) * Co + co0 + 32 * w + c4);
    *(floatx4 *)(o_tile + r * OS + 32 * w + c4) = v;
  }
  if (tid < 64) {  // pass table: lane k tabulates the passes of offset k at the position given by a wave scan
    int s = 0, c = 0;
    if (tid < K && tid % ksplit == (int)blockIdx.y) {
      s = bases[(int64_t)tid * (n_gran + 1) + g0];
      c = bases[(int64_t)tid * (n_gran + 1) + g1] - s;
    }
    const int npl = (c + 31) >> 5;
    int pos = npl * nch;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(pos, o);
      if (lane >= o) pos += y;
    }
    const int total = __shfl(pos, 63);
    pos -= npl * nch;
    const unsigned first = (unsigned)(((int64_t)tid * stride + s) * 8);
    for (int chn = 0; chn < nch; ++chn)
      for (int j = 0; j < npl; ++j) {
        ptab[pos][0] = first + (unsigned)j * 256u;
        ptab[pos][1] = (unsigned)(c - 32 * j) | (unsigned)chn << 8 | (unsigned)tid << 12 | (j == 0 ? 1u << 20 : 0u);
        ++pos;
      }
    if (tid < 8) {
      ptab[total + tid][0] = 0u;
      ptab[total + tid][1] = 0u;
    }
  }
  __syncthreads();  // the only barrier
  auto uni = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  struct Ent { unsigned base, meta; };  // meta & 255: pairs left (0: no such pass)
  auto entry = [&](int p) __attribute__((always_inline)) {
    const uint2 v = *(const uint2 *)ptab[p];
    return Ent{(unsigned)uni((int)v.x), (unsigned)uni((int)v.y)};
  };
  auto rem = [](const Ent &e) __attribute__((always_inline)) { return (int)(e.meta & 255u); };
  // per-lane constants
  const unsigned li8[2] = {(unsigned)li * 8u, (unsigned)(li + 16) * 8u};  // byte offset of the lane's pair in a pass
  const unsigned lq16 = (unsigned)lq * 16u, ci4 = (unsigned)Ci * 4u;
  // output-tile byte address of (row dst, this lane's planes) = dst * (OS * 4) + cbase   (mod 2^32: dst - o0 is what counts)
  const unsigned cbase = lds0 + (unsigned)(32 * w + 4 * lq) * 4u - (unsigned)o0 * (unsigned)(OS * 4);
  const unsigned wlane = (unsigned)lane * 16u;
  // v_mad_u32_u24: full rate (v_mul_lo_u32 / v_mad_u64_u32 are quarter rate); rows and sites are < 2^24 (conv_plan)
  auto mad24 = [](unsigned a, unsigned b, unsigned c) __attribute__((always_inline)) { return __umul24(a, b) + c; };
  // index-load offsets of a pass: the lane's pair, clamped to the last pair of the list (lanes past the end repeat it; their
  // contribution never reaches the output tile)
  auto index_offsets = [&](const Ent &e, unsigned (&vo)[2]) __attribute__((always_inline)) {
    const unsigned last = (unsigned)(rem(e) - 1) * 8u;  // rem == 0 (no such pass): no clamp, reads the first pairs of `rules`
#pragma unroll
    for (int i = 0; i < 2; ++i) vo[i] = min(li8[i], last);
  };
  // timing-only builds (tuning library): DBG & 1 no output-tile update, & 2 no row loads, & 4 no weight loads, & 8 no index
  // loads (any of 2 / 4 / 8: no waits either)
  constexpr bool NOWAIT = (DBG & 30) != 0;  // & 16: all loads, no waits
  auto load_idx = [&](const Ent &e, const unsigned (&vo)[2], u32x2 (&pr)[2]) __attribute__((always_inline)) {
    if (DBG & 8) return;
    const int base = (int)e.base;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(pr[i]) : "v"(vo[i]), "s"(ru_rsrc), "s"(base) : "memory");
  };
  auto load_piece = [&](floatx4 &dst, int sp, unsigned voff, int so) __attribute__((always_inline)) {
    if (DBG & 2) return;
    if (sp == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    else if (sp == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:64" : "+v"(dst) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    else if (sp == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:128" : "+v"(dst) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:192" : "+v"(dst) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
  };
  auto weight_offsets = [&](const Ent &e, int (&so)[NS16]) __attribute__((always_inline)) {
    const int k = (int)(e.meta >> 12) & 255, chn = (int)(e.meta >> 8) & 15;
    const int blk = (k * (Ci / 16) + chn * (CK / 16)) * (Co / 16) + co0 / 16 + w * COS;
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp) so[sp] = (blk + sp * (Co / 16)) * 1024;
  };
  // EXEC mask of the lanes whose pair exists: group g of a pass with `r` pairs left holds min(max(r - 16 g, 0), 16)
  auto lane_mask = [](int r, int g) __attribute__((always_inline)) {
    const int n = min(max(r - 16 * g, 0), 16);
    return (unsigned)((1u << n) - 1u) * 0x10001u;  // lanes li + 16 lq, lq = 0..3: the same 16 bits in all four quarters
  };
  // the two 16-plane fragments of one group -> output tile, lanes outside `mask` not written
  auto tile_write = [&](unsigned addr, unsigned mask, const floatx4 &d0, const floatx4 &d1) __attribute__((always_inline)) {
    if (DBG & 64)  // timing only: every lane written, EXEC never written
      asm volatile("ds_write_b128 %[a], %[d0]\n\tds_write_b128 %[a], %[d1] offset:64" : : [a] "v"(addr), [d0] "v"(d0), [d1] "v"(d1) : "memory");
    else
    asm volatile(
        "s_mov_b32 exec_lo, %[m]\n\t"
        "s_mov_b32 exec_hi, %[m]\n\t"
        "ds_write_b128 %[a], %[d0]\n\t"
        "ds_write_b128 %[a], %[d1] offset:64\n\t"
        "s_mov_b64 exec, -1"
        :
        : [a] "v"(addr), [m] "s"(__builtin_amdgcn_readfirstlane((int)mask)), [d0] "v"(d0), [d1] "v"(d1)
        : "memory");
  };
  auto tile_read = [&](unsigned addr, floatx4 (&old)[COS]) __attribute__((always_inline)) {
    const lds_floatx4 *p = (const lds_floatx4 *)(size_t)addr;
    old[0] = p[0];
    old[1] = p[4];
  };
#define SCN_ROWS6(R) "v"(R[0][0]), "v"(R[0][1]), "v"(R[0][2]), "v"(R[0][3]), "v"(R[1][0]), "v"(R[1][1]), "v"(R[1][2]), "v"(R[1][3])
#define SCN_W6(W) "v"(W[0][0]), "v"(W[0][1]), "v"(W[0][2]), "v"(W[0][3]), "v"(W[1][0]), "v"(W[1][1]), "v"(W[1][2]), "v"(W[1][3])

  Ent E0 = entry(0);
  if (rem(E0) > 0) {
    // ---- state at the top of pass p: rows(p) requested into R during pass p-1; weights(p) in (or on their way into) the
    // current weight set; indices of pass p+1 requested into pr; orc / mc = output-tile addresses and lane masks of pass p;
    // acc[1] / op1 / mp1 = result, address and mask of group 1 of pass p-1, not yet in the tile; E0..E2 = table entries
    // p..p+2; ioff = index-load offsets of pass p+2.
    u32x2 pr[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};
    unsigned ioff[2];
    floatx4 R[2][NS16], wA[COS][NS16], wB[COS][NS16], acc[2][COS];
    unsigned orc[2], op1 = lds0, mc[2], mp1 = 0u;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp) R[g][sp] = wA[g][sp] = wB[g][sp] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < COS; ++q) acc[1][q] = floatx4{0.f, 0.f, 0.f, 0.f};
    Ent E1 = entry(1), E2 = entry(2);
    int pnext = 3;
    {  // prologue (once per tile).  Load order = the steady state's: weights, indices, rows.
      int so[NS16];
      index_offsets(E0, ioff);
      load_idx(E0, ioff, pr);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(pr[0]), "v"(pr[1]) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      unsigned off[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        off[i] = mad24(INCOL ? pr[i].y : pr[i].x, ci4, lq16);
        orc[i] = mad24(INCOL ? pr[i].x : pr[i].y, (unsigned)(OS * 4), cbase);
        mc[i] = lane_mask(rem(E0), i);
      }
      weight_offsets(E0, so);
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wA[0][sp]) : "v"(wlane), "s"(wp_rsrc), "s"(so[sp]) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "+v"(wA[1][sp]) : "v"(wlane), "s"(wp_rsrc), "s"(so[sp]) : "memory");
      }
      index_offsets(E1, ioff);
      __builtin_amdgcn_sched_barrier(0);
      load_idx(E1, ioff, pr);
      const int rso = (int)((E0.meta >> 8) & 15u) * (CK * 4);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int sp = 0; sp < NS16; ++sp) load_piece(R[g][sp], sp, off[g], rso);
      __builtin_amdgcn_sched_barrier(0);
      index_offsets(E2, ioff);
    }

    // one pass: weights of the current phase in wc; a phase that starts with the NEXT pass gets its weights into wn
    // (WNEW: the NEXT pass starts a phase -- or does not exist --: this pass requests its weights, 18 loads instead of 10)
    auto pass = [&](auto wnew_tag, floatx4 (&wc)[COS][NS16], floatx4 (&wn)[COS][NS16]) __attribute__((always_inline)) {
      constexpr bool WNEW = decltype(wnew_tag)::value;
      const bool two = rem(E0) > 16;
      unsigned offn[2], orn[2], ioffn[2];
      floatx4 old[COS];
      int w_so[NS16];
      // ---- top: the indices of pass p+1, the first two row pieces and the weights have landed (6 younger loads: R0[2],
      // R0[3], R1[0..3])
      __builtin_amdgcn_sched_barrier(0);
      if (!NOWAIT) asm volatile("s_waitcnt vmcnt(6)" ::"v"(pr[0]), "v"(pr[1]), "v"(R[0][0]), "v"(R[0][1]), SCN_W6(wc) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      // vector-ALU cluster 1: addresses of pass p+1, output-tile fragment of group 1 of pass p-1
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        offn[i] = mad24(INCOL ? pr[i].y : pr[i].x, ci4, lq16);
        orn[i] = mad24(INCOL ? pr[i].x : pr[i].y, (unsigned)(OS * 4), cbase);
      }
      // (pins the conversion HERE: hipcc otherwise sinks parts of it to the end of the pass, keeping the old indices alive
      //  and turning their in-place reload into a copy)
      asm volatile("" : "+v"(offn[0]), "+v"(offn[1]), "+v"(orn[0]), "+v"(orn[1]));
      if (!(DBG & 1)) tile_read(op1, old);
      const u32x2 e3v = *(const u32x2 *)ptab[pnext];
      const int rson = (int)((E1.meta >> 8) & 15u) * (CK * 4);
      const unsigned mn0 = lane_mask(rem(E1), 0), mn1 = lane_mask(rem(E1), 1);
      if (WNEW) weight_offsets(E1, w_so);
      Ent E3 = E2;
      __builtin_amdgcn_sched_barrier(0);
      // ---- group 0.  Between its MFMAs: memory and LDS instructions only, except one vector-ALU cluster after the first
      // eight (group 1 of the previous pass -> output tile, table entry p+3 to scalars, index offsets of pass p+3)
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp) {
        if (sp > 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (NOWAIT || sp == 1) {  // (R0[1] has landed with the top wait)
          } else if (sp == 2 && WNEW) asm volatile("s_waitcnt vmcnt(15)" ::"v"(R[0][sp]) : "memory");  // (index loads not yet issued)
          else if (sp == 2) asm volatile("s_waitcnt vmcnt(7)" ::"v"(R[0][sp]) : "memory");
          else if (WNEW) asm volatile("s_waitcnt vmcnt(17)" ::"v"(R[0][sp]) : "memory");
          else asm volatile("s_waitcnt vmcnt(9)" ::"v"(R[0][sp]) : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const int q = h & 1, c = h >> 1;
          const float a = c == 0 ? wc[q][sp].x : c == 1 ? wc[q][sp].y : c == 2 ? wc[q][sp].z : wc[q][sp].w;
          const float bv = c == 0 ? R[0][sp].x : c == 1 ? R[0][sp].y : c == 2 ? R[0][sp].z : R[0][sp].w;
          if (sp == 0 && c == 0) acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[0][q], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (WNEW && sp == 1 && h < 4 && !(DBG & 4)) {  // the 8 weight loads of the next phase
            asm volatile(
                "buffer_load_dwordx4 %[w0], %[vo], %[rs], %[so] offen\n\t"
                "buffer_load_dwordx4 %[w1], %[vo], %[rs], %[so] offen offset:1024"
                : [w0] "+v"(wn[0][h]), [w1] "+v"(wn[1][h])
                : [vo] "v"(wlane), [rs] "s"(wp_rsrc), [so] "s"(w_so[h])
                : "memory");
          } else if (sp == 2 && h == 0) {
            load_idx(E2, ioff, pr);  // indices of pass p+2, in place
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 2048) {  // timing only: the two pieces that share a cache line of every row are requested together
          if (sp & 1) { load_piece(R[0][sp - 1], sp - 1, offn[0], rson); load_piece(R[0][sp], sp, offn[0], rson); }
        } else if (!(DBG & 1024)) load_piece(R[0][sp], sp, offn[0], rson);
        else if (sp > 0) load_piece(R[0][sp - 1], sp - 1, offn[0], rson);  // timing only: each reload one run later
        __builtin_amdgcn_sched_barrier(0);
        if (sp == 0) {  // vector-ALU cluster 2
          if (DBG & 1) asm volatile("" ::"v"(acc[1][0]), "v"(acc[1][1])); else tile_write(op1, mp1, old[0] + acc[1][0], old[1] + acc[1][1]);
          E3 = Ent{(unsigned)uni((int)e3v.x), (unsigned)uni((int)e3v.y)};
          index_offsets(E3, ioffn);
          asm volatile("" : "+v"(ioffn[0]), "+v"(ioffn[1]));  // (pinned here, not sunk to the end of the pass)
          __builtin_amdgcn_sched_barrier(0);
        } else if (sp == 2) {
          if (!(DBG & 1)) tile_read(orc[0], old);  // group 0 of this pass: its old fragment, added beside group 1
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- group 1 (if the pass has one).  Its row loads stay OUTSIDE the branch: a variable that both sides of a branch
      // redefine is copied at the join (hipcc), and these registers have loads in flight.
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp) {
        __builtin_amdgcn_sched_barrier(0);
        if (NOWAIT) {
        } else if (WNEW) asm volatile("s_waitcnt vmcnt(17)" ::"v"(R[1][sp]) : "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::"v"(R[1][sp]) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (two) {
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            const int q = h & 1, c = h >> 1;
            const float a = c == 0 ? wc[q][sp].x : c == 1 ? wc[q][sp].y : c == 2 ? wc[q][sp].z : wc[q][sp].w;
            const float bv = c == 0 ? R[1][sp].x : c == 1 ? R[1][sp].y : c == 2 ? R[1][sp].z : R[1][sp].w;
            if (sp == 0 && c == 0) acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[1][q], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 2048) {
          if (sp & 1) { load_piece(R[1][sp - 1], sp - 1, offn[1], rson); load_piece(R[1][sp], sp, offn[1], rson); }
        } else if (!(DBG & 1024)) load_piece(R[1][sp], sp, offn[1], rson);
        else if (sp == 0) load_piece(R[0][3], 3, offn[0], rson);
        else load_piece(R[1][sp - 1], sp - 1, offn[1], rson);
        if ((DBG & 1024) && sp == 3) load_piece(R[1][3], 3, offn[1], rson);
        __builtin_amdgcn_sched_barrier(0);
        if (sp == 0) {  // vector-ALU cluster 3: group 0 -> output tile (its MFMAs ended eight MFMAs ago, if there is a group 1)
          if (DBG & 1) asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][1])); else tile_write(orc[0], mc[0], old[0] + acc[0][0], old[1] + acc[0][1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      op1 = orc[1]; mp1 = mc[1];
      orc[0] = orn[0]; orc[1] = orn[1];
      mc[0] = mn0; mc[1] = mn1;
      ioff[0] = ioffn[0]; ioff[1] = ioffn[1];
      E0 = E1; E1 = E2; E2 = E3;
      ++pnext;
    };
    // The pass loop exists four times, in sequence: {passes inside a phase, last pass of a phase} x {role of the two weight
    // sets}; every copy is straight-line code with its own constants, none is the other side of a branch.  Past the end
    // of the table the passes are empty ones (no lane writes), so the second half may run one of those.
    auto last_of_phase = [&]() __attribute__((always_inline)) { return ((E1.meta >> 20) & 1u) != 0u || rem(E1) == 0; };
    do {
      while (!last_of_phase()) pass(std::false_type{}, wA, wB);
      pass(std::true_type{}, wA, wB);
      while (!last_of_phase()) pass(std::false_type{}, wB, wA);
      pass(std::true_type{}, wB, wA);
    } while (rem(E0) > 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::SCN_ROWS6(R), "v"(pr[0]), "v"(pr[1]), SCN_W6(wA), SCN_W6(wB) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    {  // group 1 of the last pass
      floatx4 old[COS];
      tile_read(op1, old);
      tile_write(op1, mp1, old[0] + acc[1][0], old[1] + acc[1][1]);
    }
  }
#undef SCN_ROWS6
#undef SCN_W6
  // each wave stores its own 32 planes: no barrier needed
  const int rows = (int)min((int64_t)M, n_out - o0);
  float *dst = out + (int64_t)blockIdx.y * n_out * Co;
  for (int i = lane; i < rows * 8; i += 64) {
    const int r = i >> 3, c4 = (i & 7) * 4;
    *(floatx4 *)(dst + (o0 + r) * Co + co0 + 32 * w + c4) = *(const floatx4 *)(o_tile + r * OS + 32 * w + c4);
  }
  if (colstats.out != nullptr && ksplit == 1) {  // column statistics of this wave's 32 planes: lane = (plane, half of the rows)
    const int p = lane & 31, half = lane >> 5;
    float cs = 0.f, css = 0.f;
    for (int r = half * (M / 2); r < min(rows, half * (M / 2) + M / 2); ++r) {
      const float v = o_tile[r * OS + 32 * w + p];
      cs += v;
      css += v * v;
    }
    cs += __shfl_xor(cs, 32);
    css += __shfl_xor(css, 32);
    if (half == 0) {
      colstats.out[((int64_t)tile * 2 + 0) * Co + co0 + 32 * w + p] = cs;
      colstats.out[((int64_t)tile * 2 + 1) * Co + co0 + 32 * w + p] = css;
    }
  }
}

// ------------------------------ forward, barrier-free, ONE wave per tile (v5) ---
// k_conv_tile4 with the other division of labour: the workgroup is a single wave that owns all 64 output planes of
// the 128-row tile (four 16-plane slices), so every gathered row is fetched once instead of once per wave, and a pass
// of 32 pairs carries 128 MFMAs instead of 64 behind the same 8 row loads, 2 index loads and scalar bookkeeping (the
// weight loads and the output-tile traffic per MFMA stay as they were).  Occupancy is one wave per SIMD (four 39 KB
// tiles per CU, as before): everything that hid behind the partner wave has to hide behind the wave's own MFMAs.
// Same pass table, same load order (16 weight, 2 index, 8 row), same hand-counted waits as k_conv_tile4.
// MEASURED SLOWER (tools/ab_tile5.py, bit-identical outputs): 84 / 86 / 87 TFLOP/s against 94 / 99 / 98 on L0 128->64,
// L1 128->128, L1 256->128; train step 54.3 against 51.4 ms.  One wave per SIMD cannot hide what two hide for each
// other, even with half the gathers.  Kept in the tuning build only (tuning key 9) so that the A/B can be repeated.
#ifdef SCN_TUNING
template <int DBG = 0>
__global__ void __launch_bounds__(64)
    k_conv_tile5(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ Wp,
                 const float *__restrict__ bias, const float *__restrict__ residual, StatsArg colstats, const int2 *__restrict__ rules,
                 const int32_t *__restrict__ bases, int K, int64_t stride, int in_col,
                 int64_t n_out, int n_gran, int Ci, int Co, int ntiles, int ncb,
                 unsigned in_bytes, unsigned wp_bytes, int ksplit) {
  constexpr int M = 128, COB = 64, OS = COB + 4, CK = 64, NS16 = 4, COS = 4;
  constexpr int TMAX = KMAX * 4 * SCN_TILE4_MAX_CHUNKS + 8;  // passes per tile: <= 4 per (offset, chunk); + sentinels
  __shared__ __attribute__((aligned(16))) float o_tile[(M + 1) * OS];  // row M: sink for padded pairs
  __shared__ __attribute__((aligned(8))) unsigned ptab[TMAX][2];  // {byte offset of the pass's first pair in `rules`,
                                                                  //  pairs left in the list | chunk<<8 | k<<12 | first<<20}
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  constexpr int w = 0;  // one wave per workgroup: it owns all 64 planes of the tile
  const int li = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Wp, 0, (int)wp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ru_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)rules, 0, (int)(K * stride * 8), 0x00020000);

  int lin;
  {
    // Workgroup b runs on XCD b % 8.  Each XCD takes CHUNKS of consecutive tiles (neighbouring tiles gather
    // overlapping rows: shared L2), dealt round robin so that density differences along the row order (whole scenes)
    // do not leave some XCDs with more work than others; the ragged end falls back to one contiguous range each.
    const int total = ntiles * ncb;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int CH = 64 * ncb;
    const int full = (DBG & 512) ? 0 : (total / (8 * CH)) * (8 * CH);  // part that divides into rounds of 8 chunks
    if (bid < full) {
      lin = ((j / CH) * 8 + xcd) * CH + j % CH;
    } else {
      const int rest = total - full, b2 = bid - full, x2 = b2 & 7, j2 = b2 >> 3;
      const int q = rest >> 3, r = rest & 7;
      lin = full + (x2 < r ? x2 * (q + 1) : r * (q + 1) + (x2 - r) * q) + j2;
    }
  }
  const int tile = lin / ncb;
  const int co0 = (lin % ncb) * COB;
  const int64_t o0 = (int64_t)tile * M;
  const int g0 = tile * (M / SCN_TILE_GRANULE);
  const int g1 = min(g0 + M / SCN_TILE_GRANULE, n_gran);
  const int nch = Ci / CK;

  for (int i = lane; i < (M + 1) * 16; i += 64) {
    const int r = i >> 4, c4 = (i & 15) * 4;
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
    if (bias && blockIdx.y == 0) v = *(const floatx4 *)(bias + co0 + c4);
    if (residual && blockIdx.y == 0 && r < M && o0 + r < n_out) v += *(const floatx4 *)(residual + (o0 + r) * Co + co0 + c4);
    *(floatx4 *)(o_tile + r * OS + c4) = v;
  }
  if (tid < 64) {  // pass table: lane k tabulates the passes of offset k at the position given by a wave scan
    int s = 0, c = 0;
    // ksplit > 1 (launches with few tiles, the coarse UNet levels): this workgroup takes the offsets k == blockIdx.y
    // (mod ksplit) and writes its partial tile to slab blockIdx.y of the launcher's workspace (k_split_reduce adds them)
    if (tid < K && tid % ksplit == (int)blockIdx.y) {
      s = bases[(int64_t)tid * (n_gran + 1) + g0];
      c = bases[(int64_t)tid * (n_gran + 1) + g1] - s;
    }
    const int npl = (c + 31) >> 5;  // passes per chunk
    int pos = npl * nch;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(pos, o);
      if (lane >= o) pos += y;
    }
    const int total = __shfl(pos, 63);
    pos -= npl * nch;
    const unsigned first = (unsigned)(((int64_t)tid * stride + s) * 8);
    for (int chn = 0; chn < nch; ++chn)
      for (int j = 0; j < npl; ++j) {
        ptab[pos][0] = first + (unsigned)j * 256u;
        ptab[pos][1] = (unsigned)(c - 32 * j) | (unsigned)chn << 8 | (unsigned)tid << 12 | (j == 0 ? 1u << 20 : 0u);
        ++pos;
      }
    if (tid < 8) {  // sentinels: empty passes
      ptab[total + tid][0] = 0u;
      ptab[total + tid][1] = 0u;
    }
  }
  __syncthreads();  // the only barrier
  auto uni = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  // offset of the loads a pass does not need (absent second group, passes past the end): the first bytes of the buffer
  // (always valid; what they fetch is multiplied by nothing or lands in the sink row).  Pointing them past the end of the
  // buffer instead (DBG & 8192: the range check returns 0 without touching memory) is 0.5-2 % SLOWER.
  constexpr unsigned OOB = (DBG & 8192) ? 0x80000000u : 0u;
  struct Ent { unsigned base, meta; };  // meta & 255: pairs left (0: no such pass)
  auto entry = [&](int p) __attribute__((always_inline)) {
    const uint2 v = *(const uint2 *)ptab[p];
    return Ent{(unsigned)uni((int)v.x), (unsigned)uni((int)v.y)};
  };
  auto rem = [](const Ent &e) __attribute__((always_inline)) { return (int)(e.meta & 255u); };
  // per-lane byte offsets of a pass's two index loads
  auto index_offsets = [&](const Ent &e, unsigned (&vo)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) vo[i] = (16 * i < rem(e) && !(DBG & 4096)) ? (unsigned)min(16 * i + li, rem(e) - 1) * 8u : OOB;
  };
  // row byte offsets / output-tile rows of a pass from its pair indices
  auto addr = [&](const Ent &e, const u32x2 (&pr)[2], unsigned (&off)[2], int (&orow)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned src = in_col ? pr[i].y : pr[i].x;
      const int dst = (int)((int64_t)(in_col ? pr[i].x : pr[i].y) - o0);
      off[i] = (16 * i < rem(e) && !(DBG & 2)) ? src * (unsigned)(Ci * 4) + (unsigned)(lq * 16) : OOB;
      orow[i] = (16 * i + li < rem(e)) ? dst * OS : M * OS;
    }
  };
  auto weight_scalars = [&](const Ent &e, int (&so)[NS16], unsigned &voff, bool &fresh) __attribute__((always_inline)) {
    const int k = (int)(e.meta >> 12) & 255, chn = (int)(e.meta >> 8) & 15;
    fresh = (e.meta >> 20) & 1u;
    const int blk = (k * (Ci / 16) + chn * (CK / 16)) * (Co / 16) + co0 / 16 + w * COS;
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp) so[sp] = (blk + sp * (Co / 16)) * 1024;
    voff = (fresh && !(DBG & 8)) ? (unsigned)lane * 16u : OOB;
  };
  auto load_idx = [&](const Ent &e, const unsigned (&vo)[2], u32x2 (&pr)[2]) __attribute__((always_inline)) {
    const int base = (int)e.base;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(pr[i]) : "v"(vo[i]), "s"(ru_rsrc), "s"(base) : "memory");
  };
  auto load_rows = [&](int i, unsigned voff, int so, floatx4 (&rows)[2][NS16]) __attribute__((always_inline)) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rows[i][0]) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:64" : "=v"(rows[i][1]) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:128" : "=v"(rows[i][2]) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:192" : "=v"(rows[i][3]) : "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
  };
  auto load_weights = [&](int sp, unsigned voff, int so, floatx4 (&wr)[COS][NS16]) __attribute__((always_inline)) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(wr[0][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(wr[1][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:2048" : "=v"(wr[2][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:3072" : "=v"(wr[3][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
  };
  // the NEXT phase's weights are loaded straight into accumulation registers (buffer loads may target AGPRs on gfx950):
  // 64 registers that do not compete with the rows and the current weights for the 256 VGPRs of the wave -- with all of
  // it in VGPRs hipcc parked freshly requested (still in flight!) load targets in AGPRs by itself
  auto load_weights_a = [&](int sp, unsigned voff, int so, floatx4 (&wr)[COS][NS16]) __attribute__((always_inline)) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(wr[0][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=a"(wr[1][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:2048" : "=a"(wr[2][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:3072" : "=a"(wr[3][sp]) : "v"(voff), "s"(wp_rsrc), "s"(so) : "memory");
  };
#define SCN_ROWS(R) "v"(R[0][0]), "v"(R[0][1]), "v"(R[0][2]), "v"(R[0][3]), "v"(R[1][0]), "v"(R[1][1]), "v"(R[1][2]), "v"(R[1][3])
#define SCN_WTS(R) SCN_ROWS(R), "v"(R[2][0]), "v"(R[2][1]), "v"(R[2][2]), "v"(R[2][3]), "v"(R[3][0]), "v"(R[3][1]), "v"(R[3][2]), "v"(R[3][3])
#define SCN_WTS_A(R) "a"(R[0][0]), "a"(R[0][1]), "a"(R[0][2]), "a"(R[0][3]), "a"(R[1][0]), "a"(R[1][1]), "a"(R[1][2]), "a"(R[1][3]), \
                     "a"(R[2][0]), "a"(R[2][1]), "a"(R[2][2]), "a"(R[2][3]), "a"(R[3][0]), "a"(R[3][1]), "a"(R[3][2]), "a"(R[3][3])
  auto scatter = [&](const floatx4 (&a)[2][COS], const floatx4 (&old)[2][COS], const int (&orow)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < COS; ++q) {
        if (DBG & 1) {
          asm volatile("" ::"v"(a[i][q]));
        } else {  // absent group / padded pair: sink row, += 0 or garbage nobody reads
          *(floatx4 *)__builtin_assume_aligned(o_tile + orow[i] + 32 * w + 16 * q + 4 * lq, 16) = old[i][q] + a[i][q];
        }
      }
  };
  auto tile_rows = [&](floatx4 (&old)[2][COS], const int (&orow)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < COS; ++q)
        // (without the assumption hipcc loses the 16-byte alignment through the rotating row variables and emits two
        //  ds_read2_b32 / ds_write2_b32 per fragment instead of one ds_read_b128 / ds_write_b128)
        if (!(DBG & 1)) old[i][q] = *(const floatx4 *)__builtin_assume_aligned(o_tile + orow[i] + 32 * w + 16 * q + 4 * lq, 16);
  };

  unsigned long long seg[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
  Ent E0 = entry(0);
  if (rem(E0) > 0) {
    // ---- pipeline state at the top of pass p: rows(p) in rc, rows(p+1) in rm (arrived), weights(p) in wc, indices of
    // pass p+2 in pr (arrived); E0..E3 = table entries p..p+3; `ioff` = index-load offsets of pass p+3.
    u32x2 pr[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};
    unsigned off[2], ioff[2];
    floatx4 rA[2][NS16], rB[2][NS16], rC[2][NS16], wc[COS][NS16], wn[COS][NS16], prev[2][COS];
    int o0r[2], o1r[2], o2r[2], op[2] = {M * OS, M * OS};  // output-tile rows of pass p, p+1, p+2, p-1
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < COS; ++q) prev[i][q] = floatx4{0.f, 0.f, 0.f, 0.f};
    Ent E1 = entry(1), E2 = entry(2), E3 = entry(3);
    int pnext = 4;  // table index of the entry read during the current pass
    {  // prologue (not overlapped; once per tile)
      int so[NS16];
      unsigned voff;
      bool fresh;
      index_offsets(E0, ioff);
      load_idx(E0, ioff, pr);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(pr[0]), "v"(pr[1]) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      addr(E0, pr, off, o0r);
      weight_scalars(E0, so, voff, fresh);
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp) load_weights(sp, (DBG & 8) ? OOB : (unsigned)lane * 16u, so[sp], wc);
      load_rows(0, off[0], (int)((E0.meta >> 8) & 15u) * (CK * 4), rA);
      load_rows(1, off[1], (int)((E0.meta >> 8) & 15u) * (CK * 4), rA);
      __builtin_amdgcn_sched_barrier(0);
      index_offsets(E1, ioff);
      load_idx(E1, ioff, pr);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(pr[0]), "v"(pr[1]), SCN_ROWS(rA), SCN_WTS(wc) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      addr(E1, pr, off, o1r);
      load_rows(0, off[0], (int)((E1.meta >> 8) & 15u) * (CK * 4), rB);
      load_rows(1, off[1], (int)((E1.meta >> 8) & 15u) * (CK * 4), rB);
      __builtin_amdgcn_sched_barrier(0);
      index_offsets(E2, ioff);
      load_idx(E2, ioff, pr);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(pr[0]), "v"(pr[1]), SCN_ROWS(rB) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      index_offsets(E3, ioff);
    }

    auto body = [&](floatx4 (&rc)[2][NS16], floatx4 (&rm)[2][NS16], floatx4 (&rn)[2][NS16]) __attribute__((always_inline)) {
      const bool two = rem(E0) > 16;
      floatx4 acc[2][COS], old[2][COS];
      Ent E4 = E3;
      int w_so[NS16], rso = 0;
      unsigned w_off = OOB;
      bool w_new = false;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < COS; ++q) acc[i][q] = floatx4{0.f, 0.f, 0.f, 0.f};
      // Everything that is not an MFMA of this pass, cut into pieces of a few instructions; piece m is issued right
      // after MFMA m of the first group (16 pieces).  Measured the same as seven larger slots between groups of four
      // MFMAs (MFMA pipe busy 66 % either way, two waves per SIMD): the granularity of the interleave is not what
      // limits the kernel, the vector-memory issue of the 10-18 loads per pass is.  Load order (vmcnt!): 8 weight,
      // 2 index, 8 row.
      auto piece = [&](int m) __attribute__((always_inline)) {
        if (m == 0 || m == 1) {  // row offset / output row of group m of pass p+2 from its (arrived) pair indices
          const int i = m;
          const unsigned src = in_col ? pr[i].y : pr[i].x;
          const int dst = (int)((int64_t)(in_col ? pr[i].x : pr[i].y) - o0);
          off[i] = (16 * i < rem(E2) && !(DBG & 2) && !((DBG & 2048) && w == 1)) ? src * (unsigned)(Ci * 4) + (unsigned)(lq * 16) : OOB;
          o2r[i] = (16 * i + li < rem(E2)) ? dst * OS : M * OS;
          if (m == 1) {
            rso = (int)((E2.meta >> 8) & 15u) * (CK * 4);
            w_new = (E1.meta >> 20) & 1u;
          }
        } else if (m == 2) {     // weights of pass p+1 only when it starts a new (offset, chunk) phase: they are OLDER than
          if (w_new) weight_scalars(E1, w_so, w_off, w_new);  // the row loads, so the vmcnt(8) below is right either way
        } else if (m >= 3 && m <= 6) {
          if (w_new) load_weights_a(m - 3, w_off, w_so[m - 3], wn);
        } else if (m == 7) {
          tile_rows(old, op);
        } else if (m == 8) {     // indices of pass p+3 (pr was consumed by pieces 0-1)
          load_idx(E3, ioff, pr);
        } else if (m >= 9 && m <= 12) {  // rows of pass p+2: the long-latency gathers, in flight until the end of pass p+1
          const int i = (m - 9) >> 1;
          if ((m - 9) & 1) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:128" : "=v"(rn[i][2]) : "v"(off[i]), "s"(in_rsrc), "s"(rso) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:192" : "=v"(rn[i][3]) : "v"(off[i]), "s"(in_rsrc), "s"(rso) : "memory");
          } else {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rn[i][0]) : "v"(off[i]), "s"(in_rsrc), "s"(rso) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:64" : "=v"(rn[i][1]) : "v"(off[i]), "s"(in_rsrc), "s"(rso) : "memory");
          }
        } else if (m == 13 || m == 14) {  // read-add-write of pass p-1, one group per piece
          const int i = m - 13;
#pragma unroll
          for (int q = 0; q < COS; ++q) {
            if (DBG & 1) {
              asm volatile("" ::"v"(prev[i][q]));
            } else {
              *(floatx4 *)__builtin_assume_aligned(o_tile + op[i] + 32 * w + 16 * q + 4 * lq, 16) = old[i][q] + prev[i][q];
            }
          }
        } else if (m == 15) {    // table entry p+4
          E4 = entry(pnext);
        }
      };
      if (!(DBG & 16)) {
#pragma unroll
        for (int sp = 0; sp < NS16; ++sp) {
#pragma unroll
          for (int h = 0; h < 16; ++h) {
            const int q = h & 3, e = h >> 2;
            acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[q][sp][e], rc[0][sp][e], acc[0][q], 0, 0, 0);
            if (sp == 0) {  // the 16 pieces ride on the first 16 of this pass's 128 MFMAs
              __builtin_amdgcn_sched_barrier(0);
              if (DBG & 256) t0 = stamp();
              piece(h);
              if (DBG & 256) { t1 = stamp(); seg[h < 3 ? 0 : h < 8 ? 1 : h < 9 ? 2 : 3] += t1 - t0; }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) piece(m);
      }
      if (two && !(DBG & 16)) {
#pragma unroll
        for (int sp = 0; sp < NS16; ++sp)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < COS; ++q)
              acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[q][sp][e], rc[1][sp][e], acc[1][q], 0, 0, 0);
      }
      // rows of pass p+1 (requested one pass ago: rm), weights of pass p+1 and indices of pass p+3 (this pass); the 8 row
      // loads of pass p+2 stay in flight.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(8)" ::SCN_ROWS(rm), SCN_WTS_A(wn), "v"(pr[0]), "v"(pr[1]) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (DBG & 256) seg[4] += 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < COS; ++q) prev[i][q] = acc[i][q];
        op[i] = o0r[i]; o0r[i] = o1r[i]; o1r[i] = o2r[i];
      }
      if (w_new) {
#pragma unroll
        for (int q = 0; q < COS; ++q)
#pragma unroll
          for (int sp = 0; sp < NS16; ++sp) wc[q][sp] = wn[q][sp];
      }
      E0 = E1; E1 = E2; E2 = E3; E3 = E4;
      index_offsets(E3, ioff);
      ++pnext;
    };
    do {  // single exit: a mid-loop break makes hipcc read "undefined" scalars out of a row register.  A pass past the
          // end is an empty one (dummy loads, output rows = sink) that only flushes `prev`.
      body(rA, rB, rC);
      body(rB, rC, rA);
      body(rC, rA, rB);
    } while (rem(E0) > 0);
    // the (dummy) row prefetches of the passes past the end
    asm volatile("s_waitcnt vmcnt(0)" ::SCN_ROWS(rA), SCN_ROWS(rB), SCN_ROWS(rC) : "memory");
    {  // result of the last pass
      floatx4 old[2][COS];
      tile_rows(old, op);
      scatter(prev, old, op);
    }
  }
#undef SCN_ROWS
#undef SCN_WTS
#undef SCN_WTS_A
  if ((DBG & 256) && lane == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&g_stamps[i], seg[i]);
  const int rows = (int)min((int64_t)M, n_out - o0);
  float *dst = out + (int64_t)blockIdx.y * n_out * Co;
  for (int i = lane; i < rows * 16; i += 64) {
    const int r = i >> 4, c4 = (i & 15) * 4;
    *(floatx4 *)(dst + (o0 + r) * Co + co0 + c4) = *(const floatx4 *)(o_tile + r * OS + c4);
  }
}


#endif  // SCN_TUNING (k_conv_tile5)

// LDS table src[k][row] = input row feeding output row o0 + row through offset k, or -1: the tile's segments of all K
// lists inverted.  8 threads per offset, all offsets in parallel: two dependent global loads per granule (the segment
// bounds, then the pairs) instead of 2 K (one offset after the other: 27 x 2 L2 round trips per 32 rows).
__device__ __forceinline__ void fill_src_table(int (*src)[SCN_TILE_GRANULE], const int2 *__restrict__ rules,
                                               const int32_t *__restrict__ bases, int K, int64_t stride, int in_col,
                                               int n_gran, int g, int64_t o0) {
  const int tid = threadIdx.x;
  for (int i = tid; i < K * SCN_TILE_GRANULE; i += 256) src[i / SCN_TILE_GRANULE][i % SCN_TILE_GRANULE] = -1;
  __syncthreads();
  const int k = tid >> 3, sub = tid & 7;
  if (k < K) {
    const int s = bases[(int64_t)k * (n_gran + 1) + g];
    const int e = bases[(int64_t)k * (n_gran + 1) + g + 1];
    const int2 *rk = rules + (int64_t)k * stride;
    for (int j = s + sub; j < e; j += 8) {
      const int2 p = rk[j];
      src[k][(int)((in_col ? p.x : p.y) - o0)] = in_col ? p.y : p.x;
    }
  }
  __syncthreads();
}

// --------------------------------------------- one contribution per output row ---
// Deconvolution, the dInput of a strided convolution and NetworkInNetwork write every output row exactly ONCE:
//   out[o,:] = in[src(o),:] . W[k(o)]          (K = 8: o = fine row, src = its coarse row, k = its parity class; K = 1: identity)
// so nothing is accumulated across offsets and the output-stationary tile of the kernels above is pure overhead
// (a 128-row output tile holds ~16 pairs per offset: every pass a new phase, weights reloaded for 16 pairs).  Here the
// unit of work is 32 consecutive PAIRS of one list x 64 output planes, owned by one wave: the whole C_i reduction runs
// in registers (A = weight fragments, B = the gathered rows fetched straight from global memory in fragment shape, as
// in k_conv_tile4), the result is stored straight to the 32 output rows.  No LDS, no barrier, no atomics.  C_i is
// walked in chunks of 32 planes, the loads of chunk c+1 requested before the MFMAs of chunk c.
// NB (round 3 experiment, tuning key 14): a wave owns NB consecutive blocks of 32 pairs and multiplies them with the SAME weight
// fragments (with one block per wave a launch reads C_i x 64 weights, 16-32 KB from L2, for every 32 gathered rows, 8-16 KB
// from HBM).  NB = 2 (240 VGPRs, two waves per SIMD) measures the same as NB = 1, NB = 4 (436 VGPRs) 20 % slower: the
// weight traffic is not the limit.  NB = 1 stays.  Neither are cross-XCD re-reads of the input: giving the ncb plane blocks of
// a pair block to ONE XCD back to back (they gather the same rows) changed no launch by more than 1.5 %.  What the launch
// by launch times say instead: a wave is a chain of dependent latencies (pair indices -> rows -> MFMAs -> residual rows ->
// stores, ~5 us for 1.8 us of MFMAs at C_i = 64) and three waves per SIMD do not cover it; 16-plane chunks (120 VGPRs, four
// waves per SIMD) bought 3 % (1.818 -> 1.768 ms per step over the 20 launches) and were not kept either.
template <int NB>
__global__ void __launch_bounds__(64)
    k_conv_expand(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ Wp,
                  const float *__restrict__ bias, const float *__restrict__ residual, StatsArg colstats, const int2 *__restrict__ rules, const int32_t *__restrict__ counts,
                  int K, int64_t stride, int in_col, int Ci, int Co, int ncb, unsigned in_bytes, unsigned wp_bytes, int nblk_xcd = 0) {
  // chunks of 32 reduction planes: 2 x (8 row + 32 weight) registers in flight + 32 accumulators keep the NB = 1 wave at
  // ~150 VGPRs = three waves per SIMD (with 64-plane chunks it needed 272: one wave per SIMD, and a launch of
  // single-chunk waves -- NiN 64 -> 128 on 600k rows -- ran at a third of the speed of the tile kernel it replaced)
  constexpr int CK = 32, NS16 = 2, COS = 4, PB = 32 * NB;
  const int lane = threadIdx.x;
  const int li = lane & 15, lq = lane >> 4;
  // (list, block of PB pairs, plane block) of this wave: blocks are numbered list after list.
  // nblk_xcd > 0 (round 5): the ncb plane blocks of one pair block sit on ONE XCD, next to each other in its queue (workgroup b
  // runs on XCD b % 8) -- numbered blockIdx.x / ncb, blockIdx.x % ncb they went to ncb different XCDs and every one of them fetched
  // the block's input rows through the fabric: NiN 256 -> 128 on level 1 read its input 2.07 times, 384 -> 192 on level 2 3.2 times
  // (profiles/r05_expand_traffic_by_kind.txt; the level-0 launches, one plane block, 1.0)
  // ... and (K > 1: deconvolution, strided-convolution dInput) the pair blocks of ONE XCD are consecutive in a sequence that
  // INTERLEAVES the lists: block r of list 0, block r of list 1, ..., block r + 1 of list 0, ...  Every list is sorted by fine and
  // by coarse site, so block r of the eight parity lists gathers the same coarse neighbourhood: numbered list after list, the
  // up to eight children of a coarse row were fetched by eight launches' worth of blocks far apart (one fabric read per PAIR);
  // interleaved and on one XCD they meet in its L2.  XCDs take chunks of 64 sequence positions round robin (as k_conv_tile4's
  // tiles), the ragged end one contiguous range each.
  int pblk;
  int cbi;
  if (nblk_xcd > 0) {
    const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int jj = j / ncb;          // position among this XCD's pair blocks
    cbi = j % ncb;
    const int total = nblk_xcd, per_x = (total + 7) >> 3;
    constexpr int CH = 64;
    const int full = (total / (8 * CH)) * (8 * CH);
    if (jj >= per_x) return;
    const int b8 = jj * 8 + x;       // (this workgroup's index if the pair blocks were dealt round robin)
    if (b8 < full) {
      pblk = ((jj / CH) * 8 + x) * CH + jj % CH;
    } else {
      const int rest = total - full, j2 = jj - full / 8, q = rest >> 3, r = rest & 7;
      if (j2 >= (x < r ? q + 1 : q)) return;
      pblk = full + (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j2;
    }
  } else {
    pblk = (int)blockIdx.x / ncb;
    cbi = (int)blockIdx.x % ncb;
  }
  const int co0 = cbi * 64;
  int k = 0, cnt = 0, blk = pblk;
  if (nblk_xcd > 0 && K > 1 && K <= 8) {
    // sequence position pblk -> (list k, block blk of it): rounds r = 0, 1, ...; round r holds one block of every list that has
    // more than r blocks, in list order.  before(r) = sum_k min(nb_k, r) blocks lie in front of round r.
    int nbk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nbmax = 0;
    for (int kk = 0; kk < K && kk < 8; ++kk) {
      nbk[kk] = (counts[kk] + PB - 1) / PB;
      nbmax = max(nbmax, nbk[kk]);
    }
    auto before = [&](int r) __attribute__((always_inline)) {
      int t = 0;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t += min(nbk[kk], r);
      return t;
    };
    if (pblk >= before(nbmax)) {
      k = K;  // past the end of the lists
    } else {
      int lo = 0, hi = nbmax - 1;  // the largest r with before(r) <= pblk
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (before(mid) <= pblk) lo = mid; else hi = mid - 1;
      }
      int within = pblk - before(lo);
      for (k = 0; k < K; ++k)
        if (nbk[k] > lo && within-- == 0) break;
      blk = lo;
      cnt = counts[k];
    }
  } else {
    for (; k < K; ++k) {
      cnt = counts[k];
      const int nb = (cnt + PB - 1) / PB;
      if (blk < nb) break;
      blk -= nb;
    }
  }
  if (k >= K) {  // past the end of the lists: nothing to compute, but this block's statistics slot must read zero
    if (colstats.out != nullptr) {
      const int64_t pb = pblk;
      colstats.out[(pb * 2 + 0) * Co + co0 + lane] = 0.f;
      colstats.out[(pb * 2 + 1) * Co + co0 + lane] = 0.f;
    }
    return;
  }
  const int p0 = blk * PB;
  const int2 *rk = rules + (int64_t)k * stride;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Wp, 0, (int)wp_bytes, 0x00020000);
  constexpr int NG = 2 * NB;  // groups of 16 pairs
  unsigned roff[NG];
  int orow[NG];
  bool live[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int p = p0 + 16 * g + li;
    live[g] = p < cnt;
    const int2 pr = rk[min(p, cnt - 1)];
    roff[g] = (unsigned)(in_col ? pr.y : pr.x) * (unsigned)((colstats.ldi ? colstats.ldi : Ci) * 4) + (unsigned)(lq * 16);
    orow[g] = in_col ? pr.x : pr.y;
  }
  const int nch = Ci / CK;
  floatx4 acc[NG][COS];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int q = 0; q < COS; ++q) acc[g][q] = floatx4{0.f, 0.f, 0.f, 0.f};
  floatx4 rA[NG][NS16], wA[COS][NS16], rB[NG][NS16], wB[COS][NS16];
  auto load = [&](int c, floatx4 (&r)[NG][NS16], floatx4 (&w)[COS][NS16]) __attribute__((always_inline)) {
    // weight fragment (k, 16-plane reduction block rs, 16-plane output block cs) = 1 KiB at ((k Ci/16 + rs) Co/16 + cs) KiB
    const int blk0 = (k * (Ci / 16) + c * (CK / 16)) * (Co / 16) + co0 / 16;
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp)
#pragma unroll
      for (int q = 0; q < COS; ++q)
        w[q][sp] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
            wp_rsrc, (unsigned)lane * 16u, (blk0 + sp * (Co / 16) + q) * 1024, 0));
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp)
        r[g][sp] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
            in_rsrc, roff[g] + (unsigned)(sp * 64), c * (CK * 4), 0));
  };
  auto mma = [&](const floatx4 (&r)[NG][NS16], const floatx4 (&w)[COS][NS16]) __attribute__((always_inline)) {
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < COS; ++q)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[g][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q][sp][e], r[g][sp][e], acc[g][q], 0, 0, 0);
  };
  load(0, rA, wA);
  for (int c = 0; c < nch; c += 2) {  // Ci % 64 == 0: an even number of 32-plane chunks
    load(c + 1, rB, wB);
    mma(rA, wA);
    if (c + 2 < nch) load(c + 2, rA, wA);
    mma(rB, wB);
  }
  // D[i = plane][j = pair]: lane (li, lq) holds planes 16 q + 4 lq .. + 3 of pair li
  const int64_t ldo = colstats.ldo ? colstats.ldo : Co;
#pragma unroll
  for (int g = 0; g < NG; ++g)
    if (live[g]) {
#pragma unroll
      for (int q = 0; q < COS; ++q) {
        floatx4 v = acc[g][q];
        if (bias) v += *(const floatx4 *)(bias + co0 + 16 * q + 4 * lq);
        if (residual) v += *(const floatx4 *)(residual + (int64_t)orow[g] * Co + co0 + 16 * q + 4 * lq);
        *(floatx4 *)(out + (int64_t)orow[g] * ldo + co0 + 16 * q + 4 * lq) = v;
        acc[g][q] = v;
      }
    } else {
#pragma unroll
      for (int q = 0; q < COS; ++q) acc[g][q] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
  // column statistics of the wave's PB output rows (colstats [waves][2][Co]; waves past the end of the lists
  // wrote zeros above): sum over the 16 pairs of a group = butterfly over the 16 lanes that share lq
  if (colstats.out != nullptr) {
    const int64_t pb = pblk;
#pragma unroll
    for (int q = 0; q < COS; ++q) {
      floatx4 cs = {0.f, 0.f, 0.f, 0.f}, css = {0.f, 0.f, 0.f, 0.f};
      StatsCols sc;
      sc.init(colstats, co0 + 16 * q + 4 * lq);
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (!sc.bn || live[g]) sc.add(acc[g][q], colstats.x + (int64_t)orow[g] * Co + co0 + 16 * q + 4 * lq, cs, css);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cs[e] += __shfl_xor(cs[e], o);
          css[e] += __shfl_xor(css[e], o);
        }
      if (li == 0) {
        *(floatx4 *)(colstats.out + (pb * 2 + 0) * Co + co0 + 16 * q + 4 * lq) = cs;
        *(floatx4 *)(colstats.out + (pb * 2 + 1) * Co + co0 + 16 * q + 4 * lq) = css;
      }
    }
  }
}

// The same arithmetic as k_conv_expand<1>, one wave per block of 32 pairs x 64 planes -- but the waves are PERSISTENT (two per
// SIMD) and walk the blocks wave, wave + G, wave + 2 G, ..., and the walk is software-pipelined ACROSS blocks: the pair indices
// of the next block are requested before this block's first multiply, and its first chunk of rows and weight fragments while
// this block's last chunk multiplies.  k_conv_expand's wave was a chain of dependent round trips -- list position -> pair
// indices -> rows -> MFMAs -> shortcut rows -> stores -- of which only the MFMAs do work: 59 TFLOP/s and 1.8 TB/s, on neither
// roofline (round 3).  Bit-identical to it (same products, same order).  Tuning library only: 0-17 % faster per launch alone (the
// 600k-row launches not at all: they move 460 MB at 3.3 TB/s), but the train step is 0.3 ms slower with it -- waves that hold
// their slots for a whole launch and no LDS share CUs with the dWeight kernels of the side stream instead of leaving them room.
#ifdef SCN_TUNING
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
    k_conv_expand_p(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ Wp,
                    const float *__restrict__ bias, const float *__restrict__ residual, StatsArg colstats, const int2 *__restrict__ rules,
                    const int32_t *__restrict__ counts, int K, int64_t stride, int in_col, int Ci, int Co, int ncb, unsigned in_bytes,
                    unsigned wp_bytes, int n_items) {
  constexpr int CK = 32, NS16 = 2, COS = 4, PB = 32, NG = 2;
  const int lane = threadIdx.x;
  const int li = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Wp, 0, (int)wp_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)((colstats.ldi ? colstats.ldi : Ci) * 4);
  const int nch = Ci / CK;
  const int64_t ldo = colstats.ldo ? colstats.ldo : Co;
  struct Blk {  // one block of the walk: list k, its pairs [p0, p0 + 32), plane block co0; k < 0: past the end of the lists
    int k, cnt, co0, pb;
    unsigned roff0, roff1;  // (named members, not arrays: hipcc kept arrays of a struct that is copied in the loop in scratch)
    int orow0, orow1;
    bool live0, live1;
  };
  static_assert(NG == 2, "two groups of 16 pairs per block");
  // (list, block of 32 pairs, plane block) of item t: blocks are numbered list after list (scalar: `counts` sits in the
  //  scalar cache after the first look)
  auto locate = [&](int t, Blk &b) __attribute__((always_inline)) {
    int blk = t / ncb;
    b.pb = blk;
    b.co0 = (t % ncb) * 64;
    int k = 0, cnt = 0;
    for (; k < K; ++k) {
      cnt = counts[k];
      const int nb = (cnt + PB - 1) / PB;
      if (blk < nb) break;
      blk -= nb;
    }
    b.k = k < K ? k : -1;
    b.cnt = cnt;
    if (b.k >= 0) {
      const int2 *rk = rules + (int64_t)k * stride;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int p = blk * PB + 16 * g + li;
        const int2 pr = rk[min(p, cnt - 1)];
        const unsigned ro = (unsigned)(in_col ? pr.y : pr.x) * row_bytes + (unsigned)(lq * 16);
        const int orw = in_col ? pr.x : pr.y;
        if (g == 0) { b.live0 = p < cnt; b.roff0 = ro; b.orow0 = orw; }
        else { b.live1 = p < cnt; b.roff1 = ro; b.orow1 = orw; }
      }
    }
  };
  floatx4 acc[NG][COS];
  floatx4 rA[NG][NS16], wA[COS][NS16], rB[NG][NS16], wB[COS][NS16];
  auto load = [&](const Blk &b, int c, floatx4 (&r)[NG][NS16], floatx4 (&w)[COS][NS16]) __attribute__((always_inline)) {
    const int blk0 = (b.k * (Ci / 16) + c * (CK / 16)) * (Co / 16) + b.co0 / 16;
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp)
#pragma unroll
      for (int q = 0; q < COS; ++q)
        w[q][sp] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
            wp_rsrc, (unsigned)lane * 16u, (blk0 + sp * (Co / 16) + q) * 1024, 0));
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int sp = 0; sp < NS16; ++sp)
        r[g][sp] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
            in_rsrc, (g == 0 ? b.roff0 : b.roff1) + (unsigned)(sp * 64), c * (CK * 4), 0));
  };
  auto mma = [&](const floatx4 (&r)[NG][NS16], const floatx4 (&w)[COS][NS16]) __attribute__((always_inline)) {
#pragma unroll
    for (int sp = 0; sp < NS16; ++sp)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < COS; ++q)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[g][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q][sp][e], r[g][sp][e], acc[g][q], 0, 0, 0);
  };
  const int G = (int)gridDim.x;
  Blk cur, nxt;
  int t = (int)blockIdx.x;
  locate(t, cur);
  if (cur.k >= 0) load(cur, 0, rA, wA);
  for (; t < n_items; t += G) {
    const bool more = t + G < n_items;
    if (more) locate(t + G, nxt);  // (its index loads travel with this block's loads)
    else nxt.k = -1;
    if (cur.k >= 0) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int q = 0; q < COS; ++q) acc[g][q] = floatx4{0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < nch; c += 2) {  // Ci % 64 == 0: an even number of 32-plane chunks
        load(cur, c + 1, rB, wB);
        mma(rA, wA);
        if (c + 2 < nch) load(cur, c + 2, rA, wA);
        else if (nxt.k >= 0) load(nxt, 0, rA, wA);  // the next block's first chunk, while this block's last one multiplies
        mma(rB, wB);
      }
      // D[i = plane][j = pair]: lane (li, lq) holds planes 16 q + 4 lq .. + 3 of pair li
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (g == 0 ? cur.live0 : cur.live1) {
          const int orw = g == 0 ? cur.orow0 : cur.orow1;
#pragma unroll
          for (int q = 0; q < COS; ++q) {
            floatx4 v = acc[g][q];
            if (bias) v += *(const floatx4 *)(bias + cur.co0 + 16 * q + 4 * lq);
            if (residual) v += *(const floatx4 *)(residual + (int64_t)orw * Co + cur.co0 + 16 * q + 4 * lq);
            *(floatx4 *)(out + (int64_t)orw * ldo + cur.co0 + 16 * q + 4 * lq) = v;
            acc[g][q] = v;
          }
        } else {
#pragma unroll
          for (int q = 0; q < COS; ++q) acc[g][q] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    } else if (nxt.k >= 0) {
      load(nxt, 0, rA, wA);  // (cannot happen: blocks past the end of the lists are the LAST ones; kept for the shape of the loop)
    }
    // column statistics of the block's 32 output rows (colstats [blocks][2][Co]; a block past the end of the lists: zeros)
    if (colstats.out != nullptr) {
      if (cur.k < 0) {
        colstats.out[((int64_t)cur.pb * 2 + 0) * Co + cur.co0 + lane] = 0.f;
        colstats.out[((int64_t)cur.pb * 2 + 1) * Co + cur.co0 + lane] = 0.f;
      } else {
#pragma unroll
        for (int q = 0; q < COS; ++q) {
          floatx4 cs = {0.f, 0.f, 0.f, 0.f}, css = {0.f, 0.f, 0.f, 0.f};
          StatsCols sc;
          sc.init(colstats, cur.co0 + 16 * q + 4 * lq);
#pragma unroll
          for (int g = 0; g < NG; ++g)
            if (!sc.bn || (g == 0 ? cur.live0 : cur.live1))
              sc.add(acc[g][q], colstats.x + (int64_t)(g == 0 ? cur.orow0 : cur.orow1) * Co + cur.co0 + 16 * q + 4 * lq, cs, css);
#pragma unroll
          for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              cs[e] += __shfl_xor(cs[e], o);
              css[e] += __shfl_xor(css[e], o);
            }
          if (li == 0) {
            *(floatx4 *)(colstats.out + ((int64_t)cur.pb * 2 + 0) * Co + cur.co0 + 16 * q + 4 * lq) = cs;
            *(floatx4 *)(colstats.out + ((int64_t)cur.pb * 2 + 1) * Co + cur.co0 + 16 * q + 4 * lq) = css;
          }
        }
      }
    }
    cur = nxt;
  }
}
#endif

// ------------------------------------------------ the stem on the matrix cores ---
// C_i <= 4 (the 3 -> m stem over 27 offsets): for a granule of 32 output rows the gathered inputs ARE a dense
// [32 x K C_i] matrix (zero where a neighbour is absent) and the layer is one small GEMM with W as [K C_i x C_o]:
//   out[32 x C_o] = A[32 x 81] . W[81 x C_o]          (k_conv_generic did it with scalar FMAs: 11 TFLOP/s, 0.26 ms for 600k sites)
// Four waves per workgroup: wave w owns the 16-row block w / 2 and two 16-plane blocks; its weight fragments (21 steps of
// v_mfma_f32_16x16x4_f32 x 2 blocks = 42 registers) stay in registers while the workgroup walks granules g, g + gridDim.x, ...
// The inverse table src[k][row] of the granule is built as in k_conv_generic, the A tile goes through LDS (row stride = 2
// mod 32 floats: the 32 lanes of a half wave, (row li, step column lq), read 32 different banks).
template <int CI>
__global__ void __launch_bounds__(256)
    k_conv_stem(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ W, const float *__restrict__ bias,
                const int2 *__restrict__ rules, const int32_t *__restrict__ bases, int K, int64_t stride, int in_col,
                int64_t n_out, int n_gran, int Co) {
  constexpr int M = SCN_TILE_GRANULE;
  constexpr int NS = (KMAX * CI + 3) / 4;  // MFMA steps over the reduction (K C_i padded to a multiple of 4)
  constexpr int AS = (NS * 4 + 31) / 32 * 32 + 2;
  __shared__ int src[KMAX][M];
  __shared__ float a[M * AS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int cb = blockIdx.y * 64;
  const int rb = w >> 1, q0 = (w & 1) * 2;
  const int kk_end = K * CI;
  bool qlive[2];
  float wf[NS][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    qlive[q] = cb + 16 * (q0 + q) < Co;  // (C_o is a multiple of 16)
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const int kk = 4 * s_ + lq;  // = k C_i + c: W is [K][C_i][C_o]
      wf[s_][q] = (qlive[q] && kk < kk_end) ? W[(int64_t)kk * Co + cb + 16 * (q0 + q) + li] : 0.f;
    }
  }
  for (int g = blockIdx.x; g < n_gran; g += gridDim.x) {
    const int64_t o0 = (int64_t)g * M;
    __syncthreads();  // the previous granule is done with src[] and a[]
    fill_src_table(src, rules, bases, K, stride, in_col, n_gran, g, o0);
    for (int i = tid; i < KMAX * M; i += 256) {  // (offset k, row r): the C_i inputs of the neighbour, or zeros
      const int k = i / M, r = i % M;
      const int sidx = k < K ? src[k][r] : -1;
#pragma unroll
      for (int c = 0; c < CI; ++c) a[r * AS + k * CI + c] = sidx >= 0 ? in[(int64_t)sidx * CI + c] : 0.f;
    }
    for (int i = tid; i < (NS * 4 - KMAX * CI) * M; i += 256) a[(i % M) * AS + KMAX * CI + i / M] = 0.f;
    __syncthreads();
    floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const float bv = a[(rb * 16 + li) * AS + 4 * s_ + lq];
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s_][q], bv, acc[q], 0, 0, 0);
    }
    // D[i = plane][j = row]: lane (li, lq) holds planes 16 q + 4 lq .. + 3 of row li
    const int64_t row = o0 + rb * 16 + li;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (qlive[q] && row < n_out) {
        floatx4 v = acc[q];
        if (bias) v += *(const floatx4 *)(bias + cb + 16 * (q0 + q) + 4 * lq);
        *(floatx4 *)(out + row * Co + cb + 16 * (q0 + q) + 4 * lq) = v;
      }
  }
}

// generic fallback (any Ci, Co; scalar FMA; used by the 3 -> m stem): 32 output rows x 64 planes per
// block.  The tile's segments of all K lists are first inverted into an LDS table src[k][row] (one
// barrier), then thread (row, 8 planes) walks the offsets on its own: no per-offset barrier, the
// weights (K*Ci*64 floats) sit in LDS when they fit.
template <bool W_IN_LDS>
__global__ void __launch_bounds__(256)
    k_conv_generic(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ W,
                   const float *__restrict__ bias, const float *__restrict__ residual, StatsArg colstats, const int2 *__restrict__ rules,
                   const int32_t *__restrict__ bases, int K, int64_t stride, int in_col,
                   int64_t n_out, int n_gran, int Ci, int Co, int w_transposed, int w_reverse) {
  constexpr int M = SCN_TILE_GRANULE;
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [K][Ci][64] when W_IN_LDS
  __shared__ int src[KMAX][M];
  const int tid = threadIdx.x;
  const int cb = blockIdx.y * 64;
  // the weights are staged ONCE per workgroup, which then walks granules blockIdx.x, blockIdx.x + gridDim.x, ...
  // (one workgroup per 32 rows re-staged 20 KB of weights for every 8 KB of output: 0.43 ms for the 600k-site stem)
  if (W_IN_LDS)
    for (int i = tid; i < K * Ci * 64; i += 256) {
      const int k = i / (Ci * 64), r = i % (Ci * 64), ci = r / 64, co = cb + r % 64;
      const int ks = w_reverse ? K - 1 - k : k;
      float v = 0.f;
      if (co < Co) v = w_transposed ? W[((int64_t)ks * Co + co) * Ci + ci] : W[((int64_t)ks * Ci + ci) * Co + co];
      wl[i] = v;
    }
  for (int g = blockIdx.x; g < n_gran; g += gridDim.x) {
  const int64_t o0 = (int64_t)g * M;
  __syncthreads();  // weights staged / previous granule done with src[]
  fill_src_table(src, rules, bases, K, stride, in_col, n_gran, g, o0);
  const int r = tid >> 3, cg = (tid & 7) * 8;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    acc[c] = (bias && cb + cg + c < Co) ? bias[cb + cg + c] : 0.f;
    if (residual && o0 + r < n_out && cb + cg + c < Co) acc[c] += residual[(o0 + r) * Co + cb + cg + c];
  }
  if (W_IN_LDS && Ci <= 4) {
    // narrow inputs (the stem): the inputs of three offsets are fetched before their FMAs -- one dependent
    // LDS -> global chain per offset made this loop pure latency
    for (int k0 = 0; k0 < K; k0 += 3) {
      float xs[3][4];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = k0 + u < K ? src[k0 + u][r] : -1;
#pragma unroll
        for (int c = 0; c < 4; ++c) xs[u][c] = (i >= 0 && c < Ci) ? in[(int64_t)i * Ci + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (k0 + u < K)
#pragma unroll
          for (int ci = 0; ci < 4; ++ci)
            if (ci < Ci) {
              const float x = xs[u][ci];
              const float4 w0 = *(const float4 *)(wl + ((k0 + u) * Ci + ci) * 64 + cg);
              const float4 w1 = *(const float4 *)(wl + ((k0 + u) * Ci + ci) * 64 + cg + 4);
              acc[0] += x * w0.x; acc[1] += x * w0.y; acc[2] += x * w0.z; acc[3] += x * w0.w;
              acc[4] += x * w1.x; acc[5] += x * w1.y; acc[6] += x * w1.z; acc[7] += x * w1.w;
            }
    }
  } else
  for (int k = 0; k < K; ++k) {
    const int i = src[k][r];
    if (i < 0) continue;
    const float *ip = in + (int64_t)i * Ci;
    const int ks = w_reverse ? K - 1 - k : k;
    for (int ci = 0; ci < Ci; ++ci) {
      const float x = ip[ci];
      if (W_IN_LDS) {
        const float4 w0 = *(const float4 *)(wl + (k * Ci + ci) * 64 + cg);
        const float4 w1 = *(const float4 *)(wl + (k * Ci + ci) * 64 + cg + 4);
        acc[0] += x * w0.x; acc[1] += x * w0.y; acc[2] += x * w0.z; acc[3] += x * w0.w;
        acc[4] += x * w1.x; acc[5] += x * w1.y; acc[6] += x * w1.z; acc[7] += x * w1.w;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int co = cb + cg + c;
          if (co < Co)
            acc[c] += x * (w_transposed ? W[((int64_t)ks * Co + co) * Ci + ci] : W[((int64_t)ks * Ci + ci) * Co + co]);
        }
      }
    }
  }
  if (o0 + r < n_out) {
    float *op = out + (o0 + r) * Co + cb + cg;
    if (Co % 4 == 0 && cb + cg + 8 <= Co) {
      *(float4 *)op = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *(float4 *)(op + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (cb + cg + c < Co) op[c] = acc[c];
    }
  }
  }
}

// ---------------------------------------------------------------- dWeight ---
// Where a dWeight workgroup leaves its partial sums.  Default: float atomics into dW (as the reference,
// sparseconvnet/SCN/CUDA/Convolution.cu:749): fastest, reproducible to rounding only.  With a workspace (`part`: [slots][K Ci Co],
// zeroed by the launcher): plain stores into the workgroup's own slot, and k_dw_reduce adds the slots to dW in a FIXED
// order afterwards: parameter gradients are then bitwise reproducible run to run.
struct DwOut {
  float *dW, *part;
  int64_t elems;  // K * Ci * Co
  int ldi, ldd;   // leading dimensions (floats) of `in` / `dout` when they are column slices of wider tensors (k_dw_tile only)
  unsigned *tickets;  // k_dw_tile<.., DET = 1> only: arrival counters of the in-kernel fixed-order reduction (zero before and after)
  int grp;            // ... and its group size: 16, or 32 when a list has more than 256 slices (one list only: NetworkInNetwork)
  __device__ __forceinline__ void add(int64_t slot, int64_t idx, float v) const {
    if (part) part[slot * elems + idx] = v;
    else unsafeAtomicAdd(dW + idx, v);
  }
};
__global__ void __launch_bounds__(256)
    k_dw_reduce(const float *__restrict__ part, float *__restrict__ dW, int64_t elems4, int nslots) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < elems4; i += (int64_t)gridDim.x * 256) {
    float4 a = ((const float4 *)part)[i];
    for (int s = 1; s < nslots; ++s) {
      const float4 b = ((const float4 *)part)[(int64_t)s * elems4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float4 d = ((float4 *)dW)[i];
    d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
    ((float4 *)dW)[i] = d;
  }
}

// Few elements, many slots (the stem: 27 x 3 x 64 floats in up to 1,024 slots -- five workgroups of k_dw_reduce walked 21 MB, 0.24 ms):
// one WAVE per float4, lane l adds slots l, l + 64, ... in that order and the 64 lane sums meet in a butterfly -- an order
// fixed by lane numbers, not by timing.
__global__ void __launch_bounds__(256)
    k_dw_reduce_wide(const float *__restrict__ part, float *__restrict__ dW, int64_t elems4, int nslots) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= elems4) return;
  const int lane = threadIdx.x & 63;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = lane; s < nslots; s += 64) {
    const float4 b = ((const float4 *)part)[(int64_t)s * elems4 + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    a.x += __shfl_xor(a.x, m); a.y += __shfl_xor(a.y, m); a.z += __shfl_xor(a.z, m); a.w += __shfl_xor(a.w, m);
  }
  if (lane == 0) {
    float4 d = ((float4 *)dW)[i];
    d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
    ((float4 *)dW)[i] = d;
  }
}

__global__ void __launch_bounds__(256)
    k_dw_reduce1(const float *__restrict__ part, float *__restrict__ dW, int64_t elems, int nslots) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= elems) return;
  float a = part[i];
  for (int s = 1; s < nslots; ++s) a += part[(int64_t)s * elems + i];
  dW[i] += a;
}

// block: 4 waves as 2x2; wave owns TI x TJ 16x16 tiles; block tile CIB x COB.
// DET = 1 (round 4): bitwise reproducible WITHOUT a memset and a second launch.  Every workgroup stores its 32 TI x 32 TJ block of
// partial sums to a slot of its own (write-through), then draws a ticket; the slices of one (offset, plane block) form groups of
// DW_GROUP consecutive slices (2 DW_GROUP when a list has more than DW_GROUP^2 slices: the single list of a NetworkInNetwork layer): the last workgroup of a group to arrive adds the group's slots in slice order and leaves the sum
// in the group's first slot, the last GROUP of the (offset, plane block) to finish adds the group sums in group order and adds
// the result to dW -- two levels, so that no workgroup reads more than DW_GROUP slots (one workgroup adding all ~250 slices of a
// level-0 list would be a 60 us tail behind a 300 us launch).  The order of every addition is fixed by the slice numbers, not by
// who arrives when.  Tickets go back to zero.  The two-launch form (memset, k_dw_tile, k_dw_reduce) cost 0.8 ms per train step,
// almost all of it the 74 extra launches.
constexpr int DW_GROUP = 16;
template <int TI, int TJ, int PS = 32, int ABL = 0, int DET = 0>
__global__ void __launch_bounds__(256)
    k_dw_tile(const float *__restrict__ in, const float *__restrict__ dout, DwOut dwo,
              const int2 *__restrict__ rules, const int32_t *__restrict__ counts, int64_t stride,
              int in_col, int Ci, int Co, int chunk, unsigned in_bytes, unsigned do_bytes, int K) {
  constexpr int CIB = 32 * TI, COB = 32 * TJ;
  constexpr int IS = CIB + 16, DS = COB + 16;  // row strides: (stride % 32) == 16 -> ds_read_b32 conflict-free
  __shared__ __attribute__((aligned(16))) float in_t[PS * IS];
  __shared__ __attribute__((aligned(16))) float do_t[PS * DS];
  // Work item = (slice c of the pair lists, offset k, plane block t), t fastest, then k, then c, and each XCD takes a
  // contiguous range of them.  Slice c is the same FRACTION [c/gx, (c+1)/gx) of every list; the lists are sorted by
  // output row, so the 27 offsets of one slice gather the same neighbourhood of `in` and exactly the same rows of
  // `dout` at the same time and meet in the XCD's L2.  (With slices of the longest list as the fast grid index whole
  // lists streamed through the chip one after the other: 2.0-2.3 GB of HBM reads per launch for 0.37 GB of operands.)
  const int gx = chunk;  // number of slices
  int lin;
  {
    const int total = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int q = total >> 3, r = total & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int nci = Ci / CIB, nt = nci * (Co / COB);
  int t = lin % nt, k = (lin / nt) % K, c = lin / (nt * K);
  if (K < 0) {  // A/B (scn_set_tuning(1, 101)): slices of the longest list, slice index fastest, no XCD mapping
    K = -K;
    c = blockIdx.x % gx; k = (blockIdx.x / gx) % K; t = blockIdx.x / (gx * K);
  }
  const int count = counts[k];
  int per = ((count + gx - 1) / gx + PS - 1) / PS * PS;  // pairs per slice of THIS list, whole steps
  if (in_col >= 2) { per = (int)((stride + gx - 1) / gx + PS - 1) / PS * PS; in_col -= 2; }
  const int p0 = c * per;
  if (p0 >= count) return;
  const int p1 = min(count, p0 + per);
  const int ci0 = (t % nci) * CIB, co0 = (t / nci) * COB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = w & 1, wj = w >> 1;
  // rows are fetched with buffer loads: 32-bit byte offsets (one mul + add per load), scalar plane offset
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t do_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dout, 0, (int)do_bytes, 0x00020000);
  const int li = lane & 15, lq = lane >> 4;
  const int2 *rk = rules + (int64_t)k * stride;

  floatx4 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

  // Staging is software pipelined: the pair indices of step s+2 and the rows of step s+1 are requested into registers
  // before the MFMAs of step s.  Beside f32 MFMAs every vector-ALU instruction costs 8-12 cycles of matrix-pipe time
  // (tools/mfma_probe.py), and the first version of this loop issued one per MFMA (64-bit pair addresses, a clamp per
  // pair, rows past the end multiplied by zero).  Now: pair indices through a buffer load whose per-lane offset is a
  // constant and whose step offset is scalar (no clamp: what lies behind a list is never used), one v_mad_u32_u24 per
  // row load, and rows past the end of the chunk -- only the LAST step of a chunk can have any -- are fetched from an
  // out-of-range offset (the range check returns zeros) instead of being multiplied by zero.
  constexpr int NI = PS * (CIB / 4) / 256;  // 16-byte chunks of input rows per thread and step
  constexpr int ND = PS * (COB / 4) / 256;
  static_assert(NI >= 1 && ND >= 1, "tile too small for 256 threads");
  const __amdgpu_buffer_rsrc_t ru_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)rk, 0, (int)(stride * 8), 0x00020000);
  unsigned pi_[NI], pd_[ND];  // the input-side / the dOut-side member of the thread's pairs
  floatx4 ri[NI], rd[ND];
  unsigned pio[NI], pdo[ND], cio[NI], cdo[ND];  // per-lane constants: byte offset of the thread's pair / of its 16-byte chunk
#pragma unroll
  for (int j = 0; j < NI; ++j) { pio[j] = (unsigned)((tid + j * 256) / (CIB / 4)) * 8u; cio[j] = (unsigned)((tid + j * 256) % (CIB / 4)) * 16u; }
#pragma unroll
  for (int j = 0; j < ND; ++j) { pdo[j] = (unsigned)((tid + j * 256) / (COB / 4)) * 8u; cdo[j] = (unsigned)((tid + j * 256) % (COB / 4)) * 16u; }
  const unsigned ci4 = (unsigned)dwo.ldi * 4u, co4 = (unsigned)dwo.ldd * 4u;
  auto load_pairs = [&](int pb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j) pi_[j] = __builtin_amdgcn_raw_buffer_load_b32(ru_rsrc, pio[j], pb * 8 + 4 * in_col, 0);
#pragma unroll
    for (int j = 0; j < ND; ++j) pd_[j] = __builtin_amdgcn_raw_buffer_load_b32(ru_rsrc, pdo[j], pb * 8 + 4 - 4 * in_col, 0);
  };
  auto load_rows = [&](int pb, bool partial) __attribute__((always_inline)) {
    if (!partial) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
        ri[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
            in_rsrc, (ABL & 1) ? cio[j] : __umul24(pi_[j], ci4) + cio[j], ci0 * 4, 0));
#pragma unroll
      for (int j = 0; j < ND; ++j)
        rd[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
            do_rsrc, (ABL & 1) ? cdo[j] : __umul24(pd_[j], co4) + cdo[j], co0 * 4, 0));
    } else {  // last step of the chunk
      const unsigned left = (unsigned)(p1 - pb) * 8u;
#pragma unroll
      for (int j = 0; j < NI; ++j)
        ri[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
            in_rsrc, pio[j] < left ? __umul24(pi_[j], ci4) + cio[j] : 0xffffff00u, ci0 * 4, 0));
#pragma unroll
      for (int j = 0; j < ND; ++j)
        rd[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
            do_rsrc, pdo[j] < left ? __umul24(pd_[j], co4) + cdo[j] : 0xffffff00u, co0 * 4, 0));
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int idx = tid + j * 256;
      *(floatx4 *)(in_t + (idx / (CIB / 4)) * IS + (idx % (CIB / 4)) * 4) = ri[j];
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int idx = tid + j * 256;
      *(floatx4 *)(do_t + (idx / (COB / 4)) * DS + (idx % (COB / 4)) * 4) = rd[j];
    }
  };
  auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ps = 0; ps < PS / 4; ++ps) {
      float a[TI], b[TJ];
      // A[i = ci][k = pair]: lane (li, lq) <- in_t[4 ps + lq][ci]; B[k = pair][j = co] <- do_t[4 ps + lq][co]
#pragma unroll
      for (int t = 0; t < TI; ++t) a[t] = in_t[(4 * ps + lq) * IS + (wi * TI + t) * 16 + li];
#pragma unroll
      for (int t = 0; t < TJ; ++t) b[t] = do_t[(4 * ps + lq) * DS + (wj * TJ + t) * 16 + li];
#pragma unroll
      for (int ta = 0; ta < TI; ++ta)
#pragma unroll
        for (int tb = 0; tb < TJ; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
  };
  // steps 0 .. S-1 of PS pairs; only step S-1 can be partial.  The loop runs the steps whose successor is full (its
  // rows are requested without masks), then -- outside, so that no variable is defined on both sides of a branch inside
  // the loop (hipcc copies those at the join) -- step S-2 with the masked request for step S-1, then step S-1.
  const int S = (p1 - p0 + PS - 1) / PS;
  load_pairs(p0);
  load_rows(p0, S == 1);
  load_pairs(p0 + PS);
  int pb = p0;
  for (; pb < p0 + (S - 2) * PS; pb += PS) {
    stage();
    __syncthreads();
    load_rows(pb + PS, false);  // indices were requested one step ago
    load_pairs(pb + 2 * PS);
    compute();
    __syncthreads();
  }
  if (S >= 2) {
    stage();
    __syncthreads();
    load_rows(pb + PS, true);
    compute();
    __syncthreads();
  }
  stage();
  __syncthreads();
  compute();
  if (DET) {
    // slot of (slice c, offset k, plane block t): 256 lanes x TI TJ float4, lane-major per accumulator (coalesced 16-byte stores)
    constexpr int NA = TI * TJ;
    const int grp = dwo.grp;
    constexpr unsigned SLOT_B = 256u * NA * 16u;
    __syncthreads();  // (in_t is read by compute() above and reused for the ticket below)
    int *tk = (int *)in_t;
    const int64_t kt = (int64_t)k * nt + t;
    auto slot_rsrc = [&](int cc) __attribute__((always_inline)) {
      return __builtin_amdgcn_make_buffer_rsrc((void *)((char *)dwo.part + (((int64_t)cc * K + k) * nt + t) * (int64_t)SLOT_B), 0, (int)SLOT_B, 0x00020000);
    };
    auto put = [&](int cc, const floatx4 *v) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t rs = slot_rsrc(cc);
#pragma unroll
      for (int a = 0; a < NA; ++a)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[a].x), __float_as_uint(v[a].y), __float_as_uint(v[a].z), __float_as_uint(v[a].w)}, rs,
                                               ((unsigned)a * 256u + (unsigned)tid) * 16u, 0, 16 /* sc1 */);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // sum of `n` slots first, first + step, ... in that order (four slots' loads in flight at a time)
    constexpr int GB = NA <= 4 ? 4 : NA <= 8 ? 2 : 1;  // slots in flight (64 registers' worth of loads)
    auto gather = [&](int first, int step, int n, floatx4 *sum) __attribute__((always_inline)) {
      for (int j0 = 0; j0 < n; j0 += GB) {
        u32x4 b[GB][NA];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const __amdgpu_buffer_rsrc_t rs = slot_rsrc(first + min(j0 + u, n - 1) * step);
#pragma unroll
          for (int a = 0; a < NA; ++a) b[u][a] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((unsigned)a * 256u + (unsigned)tid) * 16u, 0, 16 /* sc1 */);
        }
#pragma unroll
        for (int u = 0; u < GB; ++u)
          if (j0 + u < n) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
              const floatx4 bf = floatx4{__uint_as_float(b[u][a].x), __uint_as_float(b[u][a].y), __uint_as_float(b[u][a].z), __uint_as_float(b[u][a].w)};
              sum[a] = (j0 + u == 0) ? bf : sum[a] + bf;
            }
          }
      }
    };
    auto arrive = [&](unsigned *word, int expect) __attribute__((always_inline)) {
      __syncthreads();  // (every wave has drained its stores)
      if (tid == 0) tk[0] = (int)__hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const bool last = __builtin_amdgcn_readfirstlane(tk[0]) == expect - 1;
      __syncthreads();
      if (last && tid == 0) __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return last;
    };
    floatx4 v[NA];
#pragma unroll
    for (int ta = 0; ta < TI; ++ta)
#pragma unroll
      for (int tb = 0; tb < TJ; ++tb) v[ta * TJ + tb] = acc[ta][tb];
    const int n_k = (count + per - 1) / per;        // slices of this list that hold pairs (all of them arrive here)
    const int g = c / grp, ng = (n_k + grp - 1) / grp;
    if (n_k > 1) {  // (a list of ONE slice -- the coarse levels -- has nothing to add up: its workgroup updates dW itself)
    put(c, v);
    if (!arrive(dwo.tickets + kt * grp + g, min(grp, n_k - g * grp))) return;
    asm volatile("" ::: "memory");
    gather(g * grp, 1, min(grp, n_k - g * grp), v);
    }
    if (ng > 1) {
      put(g * grp, v);  // (the group's first slot has been read: it now holds the group's sum)
      if (!arrive(dwo.tickets + (int64_t)K * nt * grp + kt, ng)) return;
      asm volatile("" ::: "memory");
      gather(0, grp, ng, v);
    }
    // dW += the sum: this workgroup is the only one of the launch that touches the block
#pragma unroll
    for (int ta = 0; ta < TI; ++ta)
#pragma unroll
      for (int tb = 0; tb < TJ; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = ci0 + (wi * TI + ta) * 16 + 4 * lq + r;
          const int co = co0 + (wj * TJ + tb) * 16 + li;
          dwo.dW[((int64_t)k * Ci + ci) * Co + co] += v[ta * TJ + tb][r];
        }
    return;
  }
  // D[row = ci][col = co]: lane holds rows 4 lq + r, column li
#pragma unroll
  for (int ta = 0; ta < TI; ++ta)
#pragma unroll
    for (int tb = 0; tb < TJ; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + (wi * TI + ta) * 16 + 4 * lq + r;
        const int co = co0 + (wj * TJ + tb) * 16 + li;
        dwo.add(c, ((int64_t)k * Ci + ci) * Co + co, acc[ta][tb][r]);
      }
}

// ---- dWeight without LDS staging (k_dw_direct).  dW[k][ci][co] = sum over the pairs of list k of in[pair.in][ci] *
// dout[pair.out][co] is a GEMM whose reduction index is the PAIR, so the accumulators are stationary and the operands can
// be fetched straight into MFMA fragment registers with full-row, fully coalesced 16-byte loads:
//   one k-step = 4 pairs.  Lane (li, lq) loads in[pair 4s+lq][ci0 + 4 li .. +4] and dout[pair 4s+lq][co0 + 4 li .. +4]
//   (two buffer_load_dwordx4: the 16 lanes of a quarter read one whole 256-byte row each way); component ja of the
//   first is the A operand of the MFMAs for the ci with ci % 4 == ja, component jb of the second the B operand for the
//   co with co % 4 == jb: 16 independent v_mfma_f32_16x16x4_f32 per k-step on 16 accumulators cover the 64 x 64 block.
// No LDS, no ds_read per MFMA (the staged kernel: 40 LDS instructions per 64 MFMAs), no barrier in the loop, two
// vector-ALU instructions (row offsets) per 16 MFMAs.  Rows are requested one iteration (4 k-steps) ahead into four
// register sets that are reloaded in place, pair indices two iterations ahead.  The four waves of a workgroup take the
// 16-pair chunks of the workgroup's slice round robin (they walk the same neighbourhood together) and add their
// accumulators through 16 KB of LDS at the end, so the number of float atomics per launch is that of k_dw_tile.
template <int INCOL, bool ATOMIC = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4)))
    k_dw_direct(const float *__restrict__ in, const float *__restrict__ dout, DwOut dwo,
                const int2 *__restrict__ rules, const int32_t *__restrict__ counts, int64_t stride,
                int Ci, int Co, int gx, unsigned in_bytes, unsigned do_bytes, int K) {
  __shared__ __attribute__((aligned(16))) floatx4 red[16 * 64];
  int lin;
  {  // as k_dw_tile: (plane block t, offset k, slice c), c slowest, each XCD a contiguous range
    const int total = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int q = total >> 3, r = total & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int nci = Ci / 64, nt = nci * (Co / 64);
  const int t = lin % nt, k = (lin / nt) % K, c = lin / (nt * K);
  const int count = counts[k];
  const int per = ((count + gx - 1) / gx + 63) / 64 * 64;  // pairs per slice of THIS list
  const int p0 = c * per;
  if (p0 >= count) return;
  const int p1 = min(count, p0 + per);
  const int ci0 = (t % nci) * 64, co0 = (t / nci) * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t do_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dout, 0, (int)do_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ru_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)(rules + (int64_t)k * stride), 0, (int)(stride * 8), 0x00020000);
  const unsigned ci4 = (unsigned)dwo.ldi * 4u, co4 = (unsigned)dwo.ldd * 4u, li16 = (unsigned)li * 16u, lq8 = (unsigned)lq * 8u;
  auto mad24 = [](unsigned a_, unsigned b_, unsigned c_) __attribute__((always_inline)) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a_), "s"(b_), "v"(c_));
    return r;
  };

  floatx4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

  // this wave's iterations: chunks w, w + 4, w + 8, ... of 16 pairs
  const int nchunks = (p1 - p0 + 15) >> 4;
  const int J = (nchunks - w + 3) >> 2;  // may be 0
  u32x2 ix[2][4];  // pair indices of iterations j + 1 (set (j + 1) & 1) and j + 2
  floatx4 ra[4], rb[4];
  auto chunk_base = [&](int j) __attribute__((always_inline)) { return p0 + ((j << 2) + w) * 16; };
  auto load_idx = [&](int j, u32x2 (&dst)[4]) __attribute__((always_inline)) {
    const int so = chunk_base(j) * 8;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      dst[u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(ru_rsrc, lq8 + (unsigned)u * 32u, so, 0));
  };
  // rows of k-step u of iteration j; masked: pairs at or behind p1 come from an out-of-range offset (zeros)
  auto load_rows = [&](int j, int u, const u32x2 &pr, bool masked) __attribute__((always_inline)) {
    const unsigned src = INCOL ? pr.y : pr.x, dst = INCOL ? pr.x : pr.y;
    unsigned oa = mad24(src, ci4, li16), ob = mad24(dst, co4, li16);
    if (masked) {
      const bool ok = chunk_base(j) + 4 * u + lq < p1;
      oa = ok ? oa : 0xffffff00u;
      ob = ok ? ob : 0xffffff00u;
    }
    ra[u] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, oa, ci0 * 4, 0));
    rb[u] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(do_rsrc, ob, co0 * 4, 0));
  };
  auto mfma16 = [&](int u) __attribute__((always_inline)) {
#pragma unroll
    for (int ja = 0; ja < 4; ++ja)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
        acc[ja][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][ja], rb[u][jb], acc[ja][jb], 0, 0, 0);
  };
  if (J > 0) {
    // prologue: indices of iterations 0 and 1, rows of iteration 0
    load_idx(0, ix[0]);
    load_idx(1, ix[1]);
#pragma unroll
    for (int u = 0; u < 4; ++u) load_rows(0, u, ix[0][u], true);
    load_idx(2, ix[0]);  // (ix[0] has been consumed by the row loads above: the compiler orders the waits)
    int j = 0;
    // Trips of two iterations whose prefetches (iterations j + 1 and j + 2) are not the wave's last iteration: only the
    // LAST chunk of a slice can be partial, so these need no mask.
    for (; j + 3 < J; j += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // two iterations per trip: the index sets alternate without copies
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          mfma16(u);
          // (pinned: left alone, hipcc sinks these loads to the end of the trip, right in front of their use)
          __builtin_amdgcn_sched_barrier(0);
          load_rows(j + h + 1, u, ix[(h + 1) & 1][u], false);
          if (u == 3) load_idx(j + h + 3, ix[(h + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // the last one to three iterations: masked prefetch
    for (; j < J; ++j) {
      const int par = j & 1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        mfma16(u);
        if (j + 1 < J) load_rows(j + 1, u, par ? ix[0][u] : ix[1][u], true);
      }
      if (j + 1 < J) { if (par) load_idx(j + 3, ix[0]); else load_idx(j + 3, ix[1]); }
    }
  }
  // add the four waves' accumulators through LDS (wave 3 -> 2 -> 1 -> 0), then one set of atomics per workgroup
#pragma unroll 1
  for (int src = 3; src >= 1; --src) {
    if (w == src) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) red[(a * 4 + b) * 64 + lane] = acc[a][b];
    }
    __syncthreads();
    if (w == src - 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] += red[(a * 4 + b) * 64 + lane];
    }
    __syncthreads();
  }
  if (w == 0) {
    // D of (ja, jb): lane holds rows 4 lq + r, column li, i.e. ci = ci0 + 4 (4 lq + r) + ja, co = co0 + 4 li + jb
#pragma unroll
    for (int ja = 0; ja < 4; ++ja)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t idx = ((int64_t)k * Ci + ci0 + 4 * (4 * lq + r) + ja) * Co + co0 + 4 * li;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) dwo.add(c, idx + jb, acc[ja][jb][r]);
      }
  }
}

// dW for narrow inputs (the 3 -> m stem, Ci <= 4): thread = one output plane, wave = every fourth pair of the chunk;
// the pair and the Ci input values are wave-uniform, only the dOut row is a vector load
template <int CI>
__global__ void __launch_bounds__(256)
    k_dw_narrow(const float *__restrict__ in, const float *__restrict__ dout, DwOut dwo,
                const int2 *__restrict__ rules, const int32_t *__restrict__ counts, int64_t stride,
                int in_col, int Co, int chunk) {
  const int k = blockIdx.y;
  const int count = counts[k];
  const int p0 = blockIdx.x * chunk;
  if (p0 >= count) return;
  const int p1 = min(count, p0 + chunk);
  const int2 *rk = rules + (int64_t)k * stride;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int co = lane; co < Co; co += 64) {
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.f;
    for (int p = p0 + w; p < p1; p += 16) {  // 4 independent pair -> row chains in flight per wave
      int2 pr[4];
      float d[4], x[4][CI];
#pragma unroll
      for (int u = 0; u < 4; ++u) pr[u] = rk[min(p + 4 * u, p1 - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float *ip = in + (int64_t)(in_col ? pr[u].y : pr[u].x) * CI;
#pragma unroll
        for (int c = 0; c < CI; ++c) x[u][c] = ip[c];
        d[u] = (p + 4 * u < p1) ? dout[(int64_t)(in_col ? pr[u].x : pr[u].y) * Co + co] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < CI; ++c) acc[c] += x[u][c] * d[u];
    }
#pragma unroll
    // (with a workspace: the four waves of this workgroup would store to the same slot -- wave w takes slot 4 x + w)
    for (int c = 0; c < CI; ++c) dwo.add((int64_t)blockIdx.x * 4 + w, ((int64_t)k * CI + c) * Co + co, acc[c]);
  }
}

// dW for narrow inputs, OUTPUT-ROW major (used when the caller passes the tile-base table of the dOut column): every dOut
// row is read ONCE and meets the <= K narrow input rows of its neighbours, instead of once per pair (2 GB of dOut reads for
// the 600k-site stem, 0.62 ms; 0.15 GB here).  Workgroups are persistent: per granule of 32 rows the tile's segments are
// inverted into an LDS table src[k][row] (as the generic forward kernel does), wave w takes rows 8w..8w+7, lane = output
// plane; lane k < K fetches the input row of offset k, the values reach all lanes through v_readlane.  The K x CI
// accumulators per plane live in registers over all granules; one LDS reduction over the 4 waves and one float atomic
// per (k, c, plane) and workgroup at the end.
template <int CI>
__global__ void __launch_bounds__(256)
    k_dw_rows(const float *__restrict__ in, const float *__restrict__ dout, DwOut dwo,
              const int2 *__restrict__ rules, const int32_t *__restrict__ bases, int K, int64_t stride, int in_col,
              int n_gran, int64_t n_dout, int Co) {
  constexpr int M = SCN_TILE_GRANULE;
  __shared__ int src[KMAX][M];
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int co = blockIdx.y * 64 + lane;
  float acc[KMAX][CI];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[k][c] = 0.f;
  for (int g = blockIdx.x; g < n_gran; g += gridDim.x) {
    const int64_t o0 = (int64_t)g * M;
    __syncthreads();
    fill_src_table(src, rules, bases, K, stride, in_col, n_gran, g, o0);
    // 8 rows per wave: all their loads are requested before the first FMA (8 independent LDS -> global chains)
    int sidx[8];
    float d[8], x[8][CI];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = w * 8 + u;
      sidx[u] = lane < K ? src[lane][r] : -1;
      d[u] = (o0 + r < n_dout && co < Co) ? dout[(o0 + r) * Co + co] : 0.f;
#pragma unroll
      for (int c = 0; c < CI; ++c) x[u][c] = sidx[u] >= 0 ? in[(int64_t)sidx[u] * CI + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K && __builtin_amdgcn_readlane(sidx[u], k) >= 0) {
#pragma unroll
          for (int c = 0; c < CI; ++c)
            acc[k][c] += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[u][c]), k)) * d[u];
        }
  }
  // 4 waves hold partial sums for the same planes
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      if (k >= K) continue;
      __syncthreads();
      red[w][lane] = acc[k][c];
      __syncthreads();
      if (w == 0 && co < Co)
        dwo.add(blockIdx.x, ((int64_t)k * CI + c) * Co + co, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
    }
}

// dW of the stem on the matrix cores: with the gathered inputs of a granule as the dense tile A[32 x K C_i] (k_conv_stem),
//   dW[K C_i x C_o] += A^T[K C_i x 32] . dOut[32 x C_o]
// is 8 steps of v_mfma_f32_16x16x4_f32 per (16 reduction rows of dW, 16 planes) block.  Wave w owns the planes 16 w .. 16 w + 15
// and all six 16-row blocks of dW (96 >= K C_i rows): 24 accumulator registers that live over all granules of the
// persistent workgroup; every (row of dW, plane) is owned by ONE lane, so the end is one float atomic (or one workspace
// slot) per element and workgroup, no reduction across waves.  k_dw_rows did the same sums with v_readlane + scalar FMAs
// (0.44 ms for 600k sites, 7 TFLOP/s).
template <int CI>
__global__ void __launch_bounds__(256)
    k_dw_stem(const float *__restrict__ in, const float *__restrict__ dout, DwOut dwo, const int2 *__restrict__ rules,
              const int32_t *__restrict__ bases, int K, int64_t stride, int in_col, int n_gran, int64_t n_dout, int Co) {
  constexpr int M = SCN_TILE_GRANULE;
  constexpr int NM = (KMAX * CI + 15) / 16;  // 16-row blocks of dW
  // row stride of the A tile = 18 mod 32: the MFMA reads (rows lq, lq + 1 x 16 columns per half wave) hit 30 different banks,
  // the gather's writes (32 rows, one column) 16
  constexpr int AS = (NM * 16 + 31) / 32 * 32 + 18;
  constexpr int DS = 80;  // row stride of the dOut tile (16 mod 32: rows lq and lq + 1 of a half wave on different banks)
  __shared__ int src[KMAX][M];
  __shared__ float a[M * AS];
  __shared__ __attribute__((aligned(16))) float d[M * DS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int cb = blockIdx.y * 64;
  floatx4 acc[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int g = blockIdx.x; g < n_gran; g += gridDim.x) {
    const int64_t o0 = (int64_t)g * M;
    __syncthreads();  // the previous granule is done with src[], a[] and d[]
    // the dOut rows of the granule, coalesced (requested before the dependent chain bases -> pairs -> inputs below)
    floatx4 dv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = (tid + 256 * u) * 4, r = e / 64, c = e % 64;
      dv[u] = (o0 + r < n_dout && cb + c < Co) ? *(const floatx4 *)(dout + (o0 + r) * Co + cb + c) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    fill_src_table(src, rules, bases, K, stride, in_col, n_gran, g, o0);
    for (int i = tid; i < KMAX * M; i += 256) {
      const int k = i / M, r = i % M;
      const int sidx = k < K ? src[k][r] : -1;
#pragma unroll
      for (int c = 0; c < CI; ++c) a[r * AS + k * CI + c] = sidx >= 0 ? in[(int64_t)sidx * CI + c] : 0.f;
    }
    for (int i = tid; i < (NM * 16 - KMAX * CI) * M; i += 256) a[(i % M) * AS + KMAX * CI + i / M] = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = (tid + 256 * u) * 4, r = e / 64, c = e % 64;
      *(floatx4 *)(d + r * DS + c) = dv[u];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < M / 4; ++t) {
      const float bv = d[(4 * t + lq) * DS + 16 * w + li];  // B[k = row][j = plane]
#pragma unroll
      for (int m = 0; m < NM; ++m)  // A[i = row of dW][k = row]
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(4 * t + lq) * AS + 16 * m + li], bv, acc[m], 0, 0, 0);
    }
  }
  // D[i = row of dW][j = plane]: lane (li, lq) holds rows 16 m + 4 lq .. + 3 of plane 16 w + li
  const int co = cb + 16 * w + li;
  if (co < Co)
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = 16 * m + 4 * lq + e;
        if (kk < K * CI) dwo.add(blockIdx.x, (int64_t)kk * Co + co, acc[m][e]);
      }
}

// generic dW (any Ci, Co): block = (pair chunk, k); threads stride over (ci, co)
__global__ void __launch_bounds__(256)
    k_dw_generic(const float *__restrict__ in, const float *__restrict__ dout, DwOut dwo,
                 const int2 *__restrict__ rules, const int32_t *__restrict__ counts, int64_t stride,
                 int in_col, int Ci, int Co, int chunk) {
  const int k = blockIdx.y;
  const int count = counts[k];
  const int p0 = blockIdx.x * chunk;
  if (p0 >= count) return;
  const int p1 = min(count, p0 + chunk);
  const int2 *rk = rules + (int64_t)k * stride;
  for (int e = threadIdx.x; e < Ci * Co; e += 256) {
    const int ci = e / Co, co = e % Co;
    float acc = 0.f;
    for (int p = p0; p < p1; p += 8) {  // 8 independent pair -> row load chains in flight
      int2 pr[8];
      float x[8], d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pr[u] = rk[min(p + u, p1 - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x[u] = in[(int64_t)(in_col ? pr[u].y : pr[u].x) * Ci + ci];
        d[u] = dout[(int64_t)(in_col ? pr[u].x : pr[u].y) * Co + co];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (p + u < p1) ? x[u] * d[u] : 0.f;
    }
    dwo.add(blockIdx.x, ((int64_t)k * Ci + ci) * Co + co, acc);
  }
}

}  // namespace scn
using namespace scn;

// ---- knobs.  The product library (libscn_hip.so) is built WITHOUT -DSCN_TUNING: every knob is a constant, the
// ablation / stamp instantiations of the kernels do not exist in it and scn_set_tuning / scn_debug_stamps are not
// exported.  tools/ use libscn_hip_tuning.so (same sources + -DSCN_TUNING, built next to it by occuseg_amd/build.py).
#ifdef SCN_TUNING
#define SCN_KNOB static int
#else
#define SCN_KNOB static constexpr int
#endif
SCN_KNOB g_conv_dbg = 0;       // tuning key 1: ablation / stamp build of the gather-mm kernels (timing only: wrong results)
SCN_KNOB g_dw_dbg = 0;         // tuning key 6: ablation build of the dWeight kernel (timing only)
SCN_KNOB g_dw_slices = 256;    // tuning key 5: dWeight slices over the longest list
SCN_KNOB g_conv_small_policy = 1500;  // tuning key 4: launches with few tiles keep the 128 x 64 tile and split their
                                      // offsets until there are this many workgroups (0 = the older rule, for A/B:
                                      // 64 x 32 tiles first); +16-22 % on the three coarsest UNet levels
SCN_KNOB g_conv_small_off = 0;  // tuning key 3: 1 = always use the widest tile, never split offsets (A/B runs)
SCN_KNOB g_conv_variant = 4;    // tuning key 0: 1 = first synchronous kernel, 2 = LDS-staged pipelined kernel for every
                                // shape, 4 (default) = barrier-free k_conv_tile4 where it applies, else as 2
SCN_KNOB g_conv_small_tile4 = 1;  // tuning key 7: n > 0 = launches with few tiles and at least n output rows use k_conv_tile4 with the offset
                                  // split instead of k_conv_tile2 (1 = all of them, the default since round 5; 0 = none).  Round 2 measured
                                  // this 0.35 ms per step SLOWER (the kernel of that time, partial tiles added by a second launch); with the
                                  // pass table in global memory and the tile's last workgroup adding the partial tiles (SPLIT): level 3
                                  // (7 k sites) 1-12 % faster per launch alone, levels 4 and 5 6-25 % slower alone -- and the train step
                                  // 0.55-0.8 ms SHORTER with all of them (44.6-44.7 against 45.2-45.5 ms, four runs of tools/ab_step.py
                                  // tune7=0,1,1000,4000: two waves and 35 KB of LDS per workgroup share a CU with the side stream's dWeight
                                  // workgroups where k_conv_tile2's four waves and 70 KB do not)
SCN_KNOB g_conv_tile5 = 0;      // tuning key 9: 1 = k_conv_tile5 (one wave per tile, 64 planes) where k_conv_tile4 would run
SCN_KNOB g_conv_expand = 1;     // tuning key 8: 0 = launches with one contribution per output row use the tile kernels (A/B)
SCN_KNOB g_conv_tile6 = 0;      // tuning key 11: 96 / 128 = k_conv_tile6 with that many rows per tile where k_conv_tile4 would run unsplit
SCN_KNOB g_conv_extra_lds = 0;  // tuning key 10: bytes of unused dynamic LDS added to k_conv_tile4 launches (occupancy probe)
SCN_KNOB g_dw_direct = 0;       // tuning key 12: 1 = k_dw_direct (no LDS staging) for 64-plane blocks.  Round 3: correct (all dW parity
                                // tests), 107-130 against 105-126 TFLOP/s on the six big layer shapes in isolation, but the train
                                // step is 1.8 ms SLOWER with it (50.3 against 48.4 ms, tools/ab_step.py tune12=0,1): off
SCN_KNOB g_conv_expand_nb = 1;  // tuning key 14: blocks of 32 pairs per wave of k_conv_expand (1, 2 or 4).  Measured (round 3, 20 launches
                                // per step): 1.81 / 1.83 / 2.21 ms -- the weight re-reads are not what bounds the kernel
SCN_KNOB g_conv_expand_xcd = 2;  // tuning key 27: 0 = the plane blocks of a pair block of k_conv_expand numbered blockIdx.x % ncb (eight XCDs fetch the same rows); 1 = launches of more than one plane block use the XCD-aware numbering -- and with it, where the rulebook has 2..8 lists, the interleaved list order (which changes the order of the column-statistics partials: A/Bs of 1 against 2 differ in that too); 2 = launches of ONE plane block whose rulebook has 2..8 lists as well
SCN_KNOB g_conv_expand_persist = 0;  // tuning key 21: 1 = k_conv_expand_p (persistent waves, pipelined across blocks; tuning library only).  Round 4:
                                     // bit-identical, 0-17 % faster per launch in isolation (tools/ab_expand.py), 0.3 ms SLOWER per train step
                                     // (46.73 against 46.40 ms, tools/ab_step.py tune21=0,1): off
SCN_KNOB g_conv_stem_mfma = 1;  // tuning key 15: 0 = the C_i <= 4 stem runs k_conv_generic (scalar FMAs) instead of k_conv_stem (A/B)
SCN_KNOB g_conv_gtab = 1;       // tuning key 13: 0 = k_conv_tile4 keeps its pass table in LDS (A/B)
SCN_KNOB g_conv_asm = 1;        // tuning key 28: 0 = k_conv_tile4 runs hipcc's pass loop instead of the hand-written one (A/B; bit-identical)
SCN_KNOB g_conv_sk = 1;         // tuning key 16: stream-K for the unsplit k_conv_tile4 launches (0 = one workgroup per tile, the round-3 launch)
SCN_KNOB g_conv_sk_static = 1;      // tuning key 22: 0 = stream-K launches draw their items and slices from counters (A/B)
SCN_KNOB g_conv_sk_min_loss = 30;   // tuning key 18: ... and only where one workgroup per tile would leave at least this many permille of the last
                                    // round of resident workgroups empty (2.70 rounds: 100; 2.83: 57; 4.59: 82).  Measured with the static
                                    // schedule (tools/ab_stream_k.py, profiles/r04_ab_stream_k.txt): +12 % at 2.70 rounds, +4-6 % at 2.83,
                                    // +1-3 % at 4.59; a launch of whole rounds would pay ~1-2 % for nothing.  0: every eligible launch (A/B)
SCN_KNOB g_conv_sk_min_items = 1024;  // tuning key 17: ... for launches with at least this many (tile, column block) items
SCN_KNOB g_dw_big = 0;            // tuning key 25: 1 = k_dw_tile with 128 x 128 blocks where both plane counts are multiples of 128 (A/B)
SCN_KNOB g_dw_k8 = 1;             // tuning key 24: 0 = strided rulebooks' dWeight cut into as many slices as a submanifold rulebook's (A/B)
SCN_KNOB g_dw_fused = 1;          // tuning key 23: 0 = deterministic dWeight as memset + kernel + k_dw_reduce even when scn_conv_dweight is handed dw_state (A/B)
SCN_KNOB g_conv_split_fused = 1;  // tuning key 19: 0 = the partial tiles of a split launch are added by a second launch (k_split_reduce) even
                                  // when the caller hands in a control block (A/B)
SCN_KNOB g_conv_wide = 0;       // tuning key 2: 1 = use the 8-wave 128-plane configuration when Co % 128 == 0

// Which kernel a gather-mm launch gets.  One function decides for the launcher AND for scn_conv_workspace_bytes.
struct ConvPlan {
  int kind;    // 0 generic (any planes), 1 first synchronous tile kernel, 2 k_conv_tile2, 4 k_conv_tile4
  int nw;      // waves per workgroup = 16-plane slices per workgroup (kinds 1, 2)
  int ksplit;  // kind 2: workgroups per (tile, plane block), each taking the offsets k == j (mod ksplit)
  bool huge;   // > 4 GiB of input rows or packed weights: 32-bit buffer offsets do not reach
};
static ConvPlan conv_plan(int K, int64_t stride, int64_t n_in, int64_t n_out, int Ci, int Co, int out_unique = 0) {
  ConvPlan p{0, 0, 1, false};
  if (Ci % 16 != 0 || Co % 16 != 0) return p;
  // the pipelined kernels address `in` and the packed weights with unsigned 32-bit buffer offsets
  p.huge = n_in * Ci * 4 >= ((int64_t)1 << 32) || (int64_t)K * Ci * Co * 4 >= ((int64_t)1 << 32);
  const int s16 = Co / 16;
  if (g_conv_variant == 1 || p.huge) {
    p.kind = 1;
    p.nw = s16 % 4 == 0 ? 4 : s16 % 3 == 0 ? 3 : s16 % 2 == 0 ? 2 : 1;
    return p;
  }
  // every output row appears in exactly one pair (deconvolution, strided-conv dInput, NiN): nothing to accumulate
  // across offsets -> k_conv_expand (register accumulators over the whole C_i reduction, direct stores)
  if (out_unique && g_conv_expand != 0 && g_conv_variant == 4 && Ci % 64 == 0 && Co % 64 == 0 &&
      (int64_t)K * stride * 8 < (int64_t)0x7fffffff) {
    p.kind = 5;
    p.nw = 1;
    return p;
  }
  // Tile shape: 4 waves (128 rows x 64 planes) when that still yields >= 2 workgroups per CU; the
  // coarse UNet levels (a few hundred to a few thousand sites) keep that tile and split their offsets.
  auto blocks = [&](int nw) { return ((n_out + 32 * nw - 1) / (32 * nw)) * (int64_t)(Co / (16 * nw)); };
  const int64_t want = 512;
  if (g_conv_variant == 4 && s16 % 4 == 0 && Ci % 64 == 0 && Ci / 64 <= SCN_TILE4_MAX_CHUNKS && blocks(4) >= want &&
      n_in < (1 << 24) && n_out < (1 << 24) && (int64_t)K * stride * 8 < (int64_t)0x7fffffff) {
    p.kind = 4;
    p.nw = 2;
    return p;
  }
  // few tiles (coarse UNet levels): the same barrier-free kernel with the offsets split 3-, 9- or 27-fold over workgroups,
  // as long as the pass table of one workgroup (<= 4 passes per offset and C_i chunk) fits
  if (g_conv_small_tile4 > 0 && n_out >= g_conv_small_tile4 && g_conv_variant == 4 && !g_conv_small_off && K > 1 && s16 % 4 == 0 && Ci % 64 == 0 && Ci / 64 <= 15 &&
      blocks(4) >= 8 && blocks(4) < want && n_in < (1 << 24) && (int64_t)K * stride * 8 < (int64_t)0x7fffffff) {
    int ks = 1;
    while (ks < 27 && blocks(4) * ks < g_conv_small_policy) ks *= 3;
    if (ks > K) ks = K;
    const int per_wg = (K + ks - 1) / ks;
    if (ks > 1 && per_wg * 4 * (Ci / 64) + 8 <= KMAX * 4 * SCN_TILE4_MAX_CHUNKS + 8) {
      p.kind = 4;
      p.nw = 2;
      p.ksplit = ks;
      return p;
    }
  }
  p.kind = 2;
  if (s16 % 4 == 0 && (blocks(4) >= want || g_conv_small_off || (g_conv_small_policy >= 1 && blocks(4) >= 8))) p.nw = 4;
  else if (s16 % 3 == 0 && s16 % 2 != 0) p.nw = 3;
  else if (s16 % 2 == 0 && (blocks(2) >= want || g_conv_small_off)) p.nw = 2;
  else p.nw = 1;
  // coarse UNet levels: too few tiles to fill 256 CUs and every workgroup walks all K offsets one after the other
  // -> split the offsets over up to 27 workgroups per tile; their partial tiles are combined in a fixed order
  const int64_t nblk = blocks(p.nw);
  const int64_t target = (g_conv_small_policy >= 1 && p.nw == 4) ? g_conv_small_policy : 4096;
  if (!g_conv_small_off && K > 1 && (p.nw < 4 || nblk < 512))
    while (p.ksplit < 27 && nblk * p.ksplit < target) p.ksplit *= 3;
  if (p.ksplit > K) p.ksplit = K;
  return p;
}

template <int NW, int COS, int M, int CK>
static int launch_tile(const float *in, float *out, const float *W, const float *bias, const float *residual, StatsArg colstats,
                       const int32_t *rules, const int32_t *bases, int K, int64_t stride,
                       int in_col, int64_t n_out, int Ci, int Co, hipStream_t stream) {
  const int n_gran = (int)((n_out + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
  dim3 grid((unsigned)((n_out + M - 1) / M), (unsigned)(Co / (NW * COS * 16)));
  k_conv_tile<NW, COS, M, CK><<<grid, 64 * NW, 0, stream>>>(in, out, W, bias, residual, colstats, (const int2 *)rules, bases,
                                                          K, stride, in_col, n_out, n_gran, Ci, Co);
  SCN_CHECK_LAUNCH();
  return 0;
}

template <int NW, int COS, int CK, int NWM = 1, int M_ = 0>
static int launch_tile2(const float *in, float *out, const float *Wp, const float *bias, const float *residual, StatsArg colstats,
                        const int32_t *rules, const int32_t *bases, int K, int64_t stride,
                        int in_col, int64_t n_out, int64_t n_in, int Ci, int Co, int ksplit, float *part,
                        hipStream_t stream, unsigned *tickets = nullptr) {
  constexpr int M = M_ ? M_ : 32 * NW;
  const int n_gran = (int)((n_out + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
  const int ntiles = (int)((n_out + M - 1) / M);
  const int ncb = Co / (NW * COS * 16);
  const unsigned in_bytes = (unsigned)((n_in - 1) * (colstats.ldi ? colstats.ldi : Ci) * 4 + Ci * 4), wp_bytes = (unsigned)((int64_t)K * Ci * Co * 4);
  float *dst = ksplit > 1 ? part : out;
  const dim3 grid2((unsigned)(ntiles * ncb), (unsigned)ksplit);
  // split launches: with a control block the last workgroup of a tile adds the partial tiles itself (one ticket word per
  // (tile, plane block) behind the stream-K words; slabs < 4 GiB: 32-bit buffer offsets)
  const bool fused = ksplit > 1 && tickets != nullptr && g_conv_split_fused &&
                     (int64_t)ntiles * ncb + SK_STATE_WORDS <= (int64_t)(SCN_SK_STATE_BYTES / 4) && n_out * Co * 4 < ((int64_t)1 << 31);
#define SCN_LAUNCH_T2(D)                                                                                   \
  k_conv_tile2<NW, COS, CK, D, NWM, M_><<<grid2, 64 * NW * NWM, 0, stream>>>(                             \
      in, dst, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, in_col, n_out, n_gran, Ci, Co, ntiles, ncb, ksplit, \
      in_bytes, wp_bytes, out, fused ? tickets + SK_STATE_WORDS : nullptr)
#ifdef SCN_TUNING
  if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 1) SCN_LAUNCH_T2(1);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 9) SCN_LAUNCH_T2(9);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 27) SCN_LAUNCH_T2(27);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 11) SCN_LAUNCH_T2(11);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 25) SCN_LAUNCH_T2(25);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 32) SCN_LAUNCH_T2(32);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 64) SCN_LAUNCH_T2(64);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 128) SCN_LAUNCH_T2(128);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 256) SCN_LAUNCH_T2(256);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 4) SCN_LAUNCH_T2(4);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 512) SCN_LAUNCH_T2(512);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 260) SCN_LAUNCH_T2(260);
  else if (NW == 4 && CK == 64 && NWM == 1 && g_conv_dbg == 258) SCN_LAUNCH_T2(258);
  else
#endif
    SCN_LAUNCH_T2(0);
#undef SCN_LAUNCH_T2
  SCN_CHECK_LAUNCH();
  if (ksplit > 1 && !fused) {
    const int64_t total4 = n_out * Co / 4;
    int64_t g = (total4 + 255) / 256;
    if (g > 2048) g = 2048;
    k_split_reduce<<<(unsigned)g, 256, 0, stream>>>(part, out, total4, ksplit);
    SCN_CHECK_LAUNCH();
  }
  return 0;
}

template <int M>
constexpr size_t tile6_lds_bytes() { return (size_t)(M * 68 + (KMAX * (M / 32) * SCN_TILE4_MAX_CHUNKS + 8) * 2) * 4; }

// rows per tile of a kind-4 launch
static int tile4_rows(int ksplit) { return (g_conv_tile6 == 96 && ksplit == 1) ? 96 : 128; }

// entries (8 bytes each) of one workgroup's region of k_conv_tile4's global pass table: <= 4 passes per (offset, chunk), 8
// sentinels, rounded to whole 64-byte lines (a scalar-cache line never holds entries of two workgroups)
// (in units of 8 bytes; the assembly loop's entries are 16 bytes: twice the units)
static int tile4_gtab_stride(int K, int Ci) { return 2 * ((K * 4 * (Ci / 64) + 8 + 7) / 8 * 8); }
// a split launch that adds its partial tiles itself (k_conv_tile4's SPLIT): ticket words behind the stream-K words of the control
// block, slabs addressed with 32-bit buffer offsets
static bool split_fused_ok(int64_t items, int64_t n_out, int Co) {
  return g_conv_split_fused && items + SK_STATE_WORDS <= (int64_t)(SCN_SK_STATE_BYTES / 4) && n_out * Co * 4 < ((int64_t)1 << 31);
}
// workspace of a split k_conv_tile4 launch: the ksplit slabs, then (64-byte aligned) one pass-table region per workgroup
static size_t tile4_split_slab_bytes(int ksplit, int64_t n_out, int Co) { return ((size_t)ksplit * (size_t)n_out * (size_t)Co * 4 + 63) / 64 * 64; }
static size_t tile4_split_workspace_bytes(int K, int ksplit, int64_t n_out, int Ci, int Co) {
  const size_t regions = (size_t)((n_out + 127) / 128) * (size_t)(Co / 64) * (size_t)ksplit;
  return tile4_split_slab_bytes(ksplit, n_out, Co) + regions * (size_t)tile4_gtab_stride((K + ksplit - 1) / ksplit, Ci) * 8;
}
// k_conv_expand: blocks of 32 pairs per wave, and an upper bound of the waves of a launch (one statistics slot each)
static int expand_nb() { return g_conv_expand_nb >= 4 ? 4 : g_conv_expand_nb >= 2 ? 2 : 1; }
static int64_t expand_waves(int64_t n_out, int K) { return (n_out + 32 * expand_nb() - 1) / (32 * expand_nb()) + K; }

// stream-K geometry of an unsplit k_conv_tile4 launch: persistent workgroups (four per CU), slices (a power of two, a multiple of
// 8 XCDs x SK_CH), and where the pass tables (two per workgroup) and the slabs (one 128 x 64 tile per slice) sit in the workspace
struct SkPlan {
  int groups, slices, shift, na_items;
  size_t gtab_bytes, slab_bytes;
};
static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}
// the static stream-K schedule has every workgroup wait for a slab that another one publishes: all SCN_SK_SLICES workgroups must
// be resident at once (four per CU on 256 CUs).  Checked once against the runtime's own occupancy answer; otherwise: no stream-K.
// What the answer does NOT cover (round-4 advisor): it is the occupancy of an EMPTY chip.  With part of the chip held by other
// kernels -- another stream of this process, another process -- some of the 1,024 workgroups start late, and the launch then makes
// progress only because (a) workgroups are dispatched in launch order per XCD, (b) a workgroup's waiting piece is its LAST one
// and the piece that feeds the next slice its FIRST, and (c) the seed of slice (XCD x, j) comes from (x, j - 1) -- an earlier
// workgroup of the same XCD -- except for j % SK_CH == 0, where it comes from (x - 1, j + SK_CH - 1), a LATER one: that workgroup
// becomes resident as soon as any workgroup of its XCD whose own seed is resident finishes, and such workgroups always exist
// (the chains within an XCD resolve front to back).  The wait is bounded (~1 s, word 9 of the control block) and a give-up is a
// hard error on the host (SCN.py: _sk_check at every batch's read-back).  The host avoids the situation where it can see it:
// no stream-K while a rulebook prefetch of this process is in flight (SCN.py: _prefetch_in_flight), and the default keeps
// stream-K to the forward launches, which have the chip to themselves.  The DRAWN schedule (g_conv_sk_static = 0) has no such
// dependence -- a slice waits only for a lower slice that was drawn earlier -- and is the one to use where residency cannot be
// assumed at all.
static bool sk_all_resident() {
  static int ok = -1;
  if (ok < 0) {
    int per_cu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;  // (no GPU in this process: size queries answer for an MI355X; not cached)
    const bool q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_conv_tile4<0, 0, 1, 1>, 128, 0) == hipSuccess;
    ok = (q && (int64_t)per_cu * device_cus() >= SCN_SK_SLICES && 4 * device_cus() == SCN_SK_SLICES) ? 1 : 0;
  }
  return ok == 1;
}
static bool tile4_sk_plan(int K, int64_t n_out, int Ci, int Co, SkPlan &sp) {
  if (g_conv_sk <= 0) return false;
  const int64_t items = ((n_out + 127) / 128) * (Co / 64);
  if (items < g_conv_sk_min_items) return false;
  if (Co / 64 > SCN_SK_MAX_NCB || SCN_SK_SLICES % (Co / 64) != 0) return false;  // (1, 2 or 4 column blocks)
  sp.groups = 4 * device_cus();
  if (!sk_all_resident()) return false;
  {
    const int64_t rounds_up = (items + sp.groups - 1) / sp.groups;
    if ((rounds_up * sp.groups - items) * 1000 < (int64_t)g_conv_sk_min_loss * rounds_up * sp.groups) return false;
  }
  sp.slices = SK_SLICES;
  sp.shift = 0;
  // whole items first: all but the last 1 + fraction rounds of SK_SLICES items, in whole tiles (the rule k_sk_slices built the table for)
  sp.na_items = (int)(scn_sk_whole_tiles((n_out + 127) / 128, Co / 64, g_conv_sk_static) * (Co / 64));
#ifdef SCN_TUNING
  if (g_conv_sk == 2) sp.na_items = -(int)(items / 512 * 512);
  if (g_conv_sk == 3) { sp.na_items = (int)0x80000000; sp.groups = (int)items; }
#endif
  if (sp.slices + SK_STATE_WORDS > (int)(SCN_SK_STATE_BYTES / 4)) return false;
  sp.gtab_bytes = (size_t)sp.groups * 2 * (size_t)tile4_gtab_stride(K, Ci) * 8;
  sp.slab_bytes = (size_t)sp.slices * 128 * 64 * 4;
  return true;
}

// workspace of an unsplit k_conv_tile4 launch: the pass tables, one region per workgroup; stream-K: + one slab per slice
static size_t tile4_workspace_bytes(int K, int64_t n_out, int Ci, int Co) {
  const size_t plain = (size_t)((n_out + 127) / 128) * (size_t)(Co / 64) * (size_t)tile4_gtab_stride(K, Ci) * 8;
  SkPlan sp;
  if (!tile4_sk_plan(K, n_out, Ci, Co, sp)) return plain;
  return plain > sp.gtab_bytes + sp.slab_bytes ? plain : sp.gtab_bytes + sp.slab_bytes;
}

static int launch_tile4(const float *in, float *out, const float *Wp, const float *bias, const float *residual, StatsArg colstats, const int32_t *rules,
                        const int32_t *bases, int K, int64_t stride, int in_col, int64_t n_out, int64_t n_in,
                        int Ci, int Co, int ksplit, float *part, const int32_t *tile_passes, unsigned *sk_state, hipStream_t stream) {
  const int M = tile4_rows(ksplit);
  const int n_gran = (int)((n_out + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
  const int ntiles = (int)((n_out + M - 1) / M);
  const int ncb = Co / 64;
  const unsigned in_bytes = (unsigned)((n_in - 1) * (colstats.ldi ? colstats.ldi : Ci) * 4 + Ci * 4), wp_bytes = (unsigned)((int64_t)K * Ci * Co * 4);
  float *dst4 = ksplit > 1 ? part : out;
  // unsplit launches: the workspace holds the pass tables (scn_conv_workspace_bytes)
  unsigned long long *gtab = (ksplit == 1 && g_conv_gtab) ? (unsigned long long *)part : nullptr;
  const int gstride = tile4_gtab_stride(K, Ci);
  const dim3 grid4((unsigned)(ntiles * ncb), (unsigned)ksplit);
  // split launches with a control block: the last workgroup of a tile adds the partial tiles itself (k_conv_tile4's SPLIT; one
  // ticket word per (tile, column block) behind the stream-K words), pass tables in global memory behind the slabs
  if (ksplit > 1 && sk_state != nullptr && g_conv_gtab && g_conv_dbg == 0 && g_conv_tile5 == 0 && M == 128 &&
      split_fused_ok((int64_t)ntiles * ncb, n_out, Co)) {
    unsigned long long *gtab_s = (unsigned long long *)((char *)part + tile4_split_slab_bytes(ksplit, n_out, Co));
    const int gs = tile4_gtab_stride((K + ksplit - 1) / ksplit, Ci);
#define SCN_LAUNCH_T4_SPLIT(IC, A)                                                                                                    \
  k_conv_tile4<0, IC, 1, 0, 1, A><<<grid4, 128, 0, stream>>>(in, part, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, in_col, \
                                                             n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, ksplit, gtab_s, gs, nullptr, \
                                                             sk_state + SK_STATE_WORDS, out, 0, 0)
    if (g_conv_asm) { if (in_col) SCN_LAUNCH_T4_SPLIT(1, 1); else SCN_LAUNCH_T4_SPLIT(0, 1); }
    else { if (in_col) SCN_LAUNCH_T4_SPLIT(1, 0); else SCN_LAUNCH_T4_SPLIT(0, 0); }
#undef SCN_LAUNCH_T4_SPLIT
    SCN_CHECK_LAUNCH();
    return 0;
  }
#ifdef SCN_TUNING  // k_conv_tile6: experimental (tools/ab_tile6.py), not faster than k_conv_tile4
  if (M != 128 || g_conv_tile6 == 128) {
#define SCN_LAUNCH_TILE6(MM, IC)                                                                                            \
  k_conv_tile6<MM, IC><<<grid4, 128, tile6_lds_bytes<MM>() + (size_t)g_conv_extra_lds, stream>>>(                            \
      in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, n_out, n_gran, Ci, Co, ntiles, ncb,   \
      in_bytes, wp_bytes, ksplit)
    if (M == 96) { if (in_col) SCN_LAUNCH_TILE6(96, true); else SCN_LAUNCH_TILE6(96, false); }
    else if (in_col) SCN_LAUNCH_TILE6(128, true);
    else {
#ifdef SCN_TUNING
#define SCN_LAUNCH_TILE6D(D)                                                                                                \
  k_conv_tile6<128, false, D><<<grid4, 128, tile6_lds_bytes<128>() + (size_t)g_conv_extra_lds, stream>>>(                   \
      in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, n_out, n_gran, Ci, Co, ntiles, ncb,   \
      in_bytes, wp_bytes, ksplit)
      switch (g_conv_dbg) {  // timing-only builds (tools/ablate_tile6.py)
        case 1: SCN_LAUNCH_TILE6D(1); break;
        case 3: SCN_LAUNCH_TILE6D(3); break;
        case 4: SCN_LAUNCH_TILE6D(4); break;
        case 14: SCN_LAUNCH_TILE6D(14); break;
        case 15: SCN_LAUNCH_TILE6D(15); break;
        case 16: SCN_LAUNCH_TILE6D(16); break;
        case 20: SCN_LAUNCH_TILE6D(20); break;
        case 32: SCN_LAUNCH_TILE6D(32); break;
        case 64: SCN_LAUNCH_TILE6D(64); break;
        case 96: SCN_LAUNCH_TILE6D(96); break;
        case 144: SCN_LAUNCH_TILE6D(144); break;
        case 272: SCN_LAUNCH_TILE6D(272); break;
        case 1040: SCN_LAUNCH_TILE6D(1040); break;
        case 2064: SCN_LAUNCH_TILE6D(2064); break;
        default: SCN_LAUNCH_TILE6D(0);
      }
#undef SCN_LAUNCH_TILE6D
#else
      SCN_LAUNCH_TILE6(128, false);
#endif
    }
#undef SCN_LAUNCH_TILE6
    SCN_CHECK_LAUNCH();
    return 0;
  }
#endif
#define SCN_LAUNCH_TILE4(D)                                                                                         \
  k_conv_tile4<D><<<grid4, 128, (size_t)g_conv_extra_lds, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K,                     \
                                             stride, in_col, n_out, n_gran, Ci, Co, ntiles, ncb,                    \
                                             in_bytes, wp_bytes, ksplit)
#ifdef SCN_TUNING
  if (g_conv_tile5 != 0) {  // one wave per tile, all 64 planes (k_conv_tile5): A/B only, measured slower
    k_conv_tile5<0><<<grid4, 64, 0, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, in_col, n_out,
                                             n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, ksplit);
  } else
#endif
  {
#ifdef SCN_TUNING
  switch (g_conv_dbg) {  // ablation builds for tools/bench_conv.py and tools/stamps.py
    case 1: SCN_LAUNCH_TILE4(1); break;     // no scatter
    case 2: SCN_LAUNCH_TILE4(3); break;     // + no gather
    case 8: SCN_LAUNCH_TILE4(11); break;    // + no weights: loop skeleton + MFMAs
    case 256: SCN_LAUNCH_TILE4(256); break;  // stamps
    case 512: SCN_LAUNCH_TILE4(512); break;  // A/B: one contiguous tile range per XCD
    case 2048: SCN_LAUNCH_TILE4(2048); break;  // timing only: wave 1 gathers nothing
    case 4096: SCN_LAUNCH_TILE4(4096 + 3); break;  // timing only: no gathers, no tile update, no index loads
    case 4097: SCN_LAUNCH_TILE4(4096 + 11); break;  // ... and no weight loads
    case 8192: SCN_LAUNCH_TILE4(8192); break;  // A/B: unneeded loads point out of range instead of reading offset 0
    default:
#define SCN_LAUNCH_T4_SK(IC, A)                                                                                                       \
  k_conv_tile4<0, IC, 1, 1, 0, A><<<sp.groups, 128, (size_t)g_conv_extra_lds, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, in_col, \
                                                                 n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, 1, gtab, gstride, tile_passes, \
                                                                 sk_state, slabs, sp.na_items, g_conv_sk_static)
#define SCN_LAUNCH_T4_GT(IC, A)                                                                                                       \
  k_conv_tile4<0, IC, 1, 0, 0, A><<<grid4, 128, (size_t)g_conv_extra_lds, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, in_col,  \
                                                             n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, ksplit, gtab, gstride)
      SkPlan sp;
      if (gtab != nullptr && tile_passes != nullptr && sk_state != nullptr && tile4_sk_plan(K, n_out, Ci, Co, sp)) {
        float *slabs = (float *)((char *)part + sp.gtab_bytes);
        if (g_conv_asm) { if (in_col) SCN_LAUNCH_T4_SK(1, 1); else SCN_LAUNCH_T4_SK(0, 1); }
        else { if (in_col) SCN_LAUNCH_T4_SK(1, 0); else SCN_LAUNCH_T4_SK(0, 0); }
      } else if (gtab != nullptr) {
        if (g_conv_asm >= 2 && !in_col) {
          switch (g_conv_asm) {
            case 2: SCN_LAUNCH_T4_GT(0, 2); break;
            case 3: SCN_LAUNCH_T4_GT(0, 3); break;
            case 4: SCN_LAUNCH_T4_GT(0, 4); break;
            case 5: SCN_LAUNCH_T4_GT(0, 5); break;
            case 6: SCN_LAUNCH_T4_GT(0, 6); break;
            default: SCN_LAUNCH_T4_GT(0, 9); break;
          }
        } else if (g_conv_asm) { if (in_col) SCN_LAUNCH_T4_GT(1, 1); else SCN_LAUNCH_T4_GT(0, 1); }
        else { if (in_col) SCN_LAUNCH_T4_GT(1, 0); else SCN_LAUNCH_T4_GT(0, 0); }
      }
#undef SCN_LAUNCH_T4_SK
#undef SCN_LAUNCH_T4_GT
      else if (in_col)
        k_conv_tile4<0, 1><<<grid4, 128, (size_t)g_conv_extra_lds, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K,
                                                                              stride, in_col, n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, ksplit);
      else
        k_conv_tile4<0, 0><<<grid4, 128, (size_t)g_conv_extra_lds, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K,
                                                                              stride, in_col, n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, ksplit);
  }
#else
#define SCN_LAUNCH_T4_SK(IC, A)                                                                                                       \
  k_conv_tile4<0, IC, 1, 1, 0, A><<<sp.groups, 128, 0, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, in_col, \
                                                                 n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, 1, gtab, gstride, tile_passes, \
                                                                 sk_state, slabs, sp.na_items, g_conv_sk_static)
#define SCN_LAUNCH_T4_GT(IC, A)                                                                                                       \
  k_conv_tile4<0, IC, 1, 0, 0, A><<<grid4, 128, 0, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K, stride, in_col,  \
                                                             n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, ksplit, gtab, gstride)
  SkPlan sp;
  if (gtab != nullptr && tile_passes != nullptr && sk_state != nullptr && tile4_sk_plan(K, n_out, Ci, Co, sp)) {
    // stream-K: persistent workgroups, equal slices of the pass sequence (see k_conv_tile4's SK)
    float *slabs = (float *)((char *)part + sp.gtab_bytes);
    if (g_conv_asm) { if (in_col) SCN_LAUNCH_T4_SK(1, 1); else SCN_LAUNCH_T4_SK(0, 1); }
    else { if (in_col) SCN_LAUNCH_T4_SK(1, 0); else SCN_LAUNCH_T4_SK(0, 0); }
  } else if (gtab != nullptr) {  // pass table in global memory, read with scalar loads (see k_conv_tile4's GT)
    if (g_conv_asm) { if (in_col) SCN_LAUNCH_T4_GT(1, 1); else SCN_LAUNCH_T4_GT(0, 1); }
    else { if (in_col) SCN_LAUNCH_T4_GT(1, 0); else SCN_LAUNCH_T4_GT(0, 0); }
#undef SCN_LAUNCH_T4_SK
#undef SCN_LAUNCH_T4_GT
  } else if (in_col)
    k_conv_tile4<0, 1><<<grid4, 128, (size_t)g_conv_extra_lds, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K,
                                                                          stride, in_col, n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, ksplit);
  else
    k_conv_tile4<0, 0><<<grid4, 128, (size_t)g_conv_extra_lds, stream>>>(in, dst4, Wp, bias, residual, colstats, (const int2 *)rules, bases, K,
                                                                          stride, in_col, n_out, n_gran, Ci, Co, ntiles, ncb, in_bytes, wp_bytes, ksplit);
#endif
  }
#undef SCN_LAUNCH_TILE4
  SCN_CHECK_LAUNCH();
  if (ksplit > 1) {
    const int64_t total4 = n_out * Co / 4;
    int64_t g = (total4 + 255) / 256;
    if (g > 2048) g = 2048;
    k_split_reduce<<<(unsigned)g, 256, 0, stream>>>(part, out, total4, ksplit);
    SCN_CHECK_LAUNCH();
  }
  return 0;
}

template <int NW>
static int launch_tile2_ck(const float *in, float *out, const float *Wp, const float *bias, const float *residual, StatsArg colstats,
                           const int32_t *rules, const int32_t *bases, int K, int64_t stride,
                           int in_col, int64_t n_out, int64_t n_in, int Ci, int Co, int ksplit, float *part,
                           hipStream_t stream, unsigned *tickets) {
#ifdef SCN_TUNING
  if (NW == 4 && g_conv_wide && Ci % 64 == 0 && Co % 128 == 0 && ksplit == 1)
    return launch_tile2<4, 2, 64, 2, 128>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, 1, part, stream);
#endif
  if (Ci % 64 == 0) return launch_tile2<NW, 1, 64>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, ksplit, part, stream, tickets);
  if (Ci % 32 == 0) return launch_tile2<NW, 1, 32>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, ksplit, part, stream, tickets);
  return launch_tile2<NW, 1, 16>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, ksplit, part, stream, tickets);
}

extern "C" {

#ifdef SCN_TUNING
// debugging aid: read (and clear) the per-segment cycle sums of the diagnostic kernel build
int scn_debug_stamps(unsigned long long *out_host) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  SCN_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(z)));
  SCN_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)));
  return 0;
}

int scn_set_tuning(int key, int value) {
  if (key == 0 && (value == 1 || value == 2 || value == 4)) { g_conv_variant = value; return 0; }
  if (key == 1) { g_conv_dbg = value; return 0; }
  if (key == 2) { g_conv_wide = value; return 0; }
  if (key == 3) { g_conv_small_off = value; return 0; }
  if (key == 4) { g_conv_small_policy = value; return 0; }
  if (key == 5 && value >= 1) { g_dw_slices = value; return 0; }
  if (key == 6) { g_dw_dbg = value; return 0; }
  if (key == 7) { g_conv_small_tile4 = value; return 0; }
  if (key == 8) { g_conv_expand = value; return 0; }
  if (key == 9) { g_conv_tile5 = value; return 0; }
  if (key == 11 && (value == 0 || value == 96 || value == 128)) { g_conv_tile6 = value; return 0; }
  if (key == 12) { g_dw_direct = value; return 0; }
  if (key == 13) { g_conv_gtab = value; return 0; }
  if (key == 28) { g_conv_asm = value; return 0; }
  if (key == 14) { g_conv_expand_nb = value; return 0; }
  if (key == 15) { g_conv_stem_mfma = value; return 0; }
  if (key == 16 && value >= 0 && value <= 3) { g_conv_sk = value; return 0; }
  if (key == 17 && value >= 1) { g_conv_sk_min_items = value; return 0; }
  if (key == 18 && value >= 0 && value <= 1000) { g_conv_sk_min_loss = value; return 0; }
  if (key == 19) { g_conv_split_fused = value; return 0; }
  if (key == 23) { g_dw_fused = value; return 0; }
  if (key == 24) { g_dw_k8 = value; return 0; }
  if (key == 25) { g_dw_big = value; return 0; }
  if (key == 27) { g_conv_expand_xcd = value; return 0; }
  if (key == 21) { g_conv_expand_persist = value; return 0; }
  if (key == 22) { g_conv_sk_static = value; return 0; }
  if (key == 20) { SCN_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_sk_prio_dev), &value, sizeof(int))); return 0; }
  if (key == 10 && value >= 0 && value <= 28000) { g_conv_extra_lds = value; return 0; }
  return fail(2, "scn_set_tuning: unknown key/value%s", "");
}
#endif

size_t scn_conv_wpack_bytes(int K, int Ci, int Co) { return (size_t)K * Ci * Co * 4; }

size_t scn_conv_workspace_bytes(int K, int64_t stride, int64_t n_in, int64_t n_out, int Ci, int Co, int out_unique) {
  if (K < 1 || n_out <= 0 || Ci <= 0 || Co <= 0) return 0;
  const ConvPlan p = conv_plan(K, stride, n_in, n_out, Ci, Co, out_unique);
  if (p.kind == 4 && p.ksplit == 1) return tile4_workspace_bytes(K, n_out, Ci, Co);
  if (p.kind == 4) return tile4_split_workspace_bytes(K, p.ksplit, n_out, Ci, Co);
  return p.ksplit > 1 ? (size_t)p.ksplit * (size_t)n_out * (size_t)Co * 4 : 0;
}

int64_t scn_conv_stats_blocks(int K, int64_t stride, int64_t n_in, int64_t n_out, int Ci, int Co, int out_unique) {
  if (K < 1 || n_out <= 0 || Ci <= 0 || Co <= 0) return 0;
  const ConvPlan p = conv_plan(K, stride, n_in, n_out, Ci, Co, out_unique);
  if (p.kind == 5) return expand_waves(n_out, K);
  if (p.kind != 2 && p.kind != 4) return 0;  // generic / fallback kernels do not emit statistics
  const int M = p.kind == 4 ? tile4_rows(p.ksplit) : 32 * p.nw;
  if (p.ksplit > 1) {
    // split launches (coarse levels): the workgroup that adds a tile's partial tiles (k_conv_tile2 with a control block) has
    // the finished tile in LDS and emits its sums like an unsplit launch.  Tensors of <= 2048 rows: none -- their batch norms
    // run as ONE launch (k_bn_small_fwd / _bwd), which a hand-over would turn into two.  (A reduction kernel that also
    // reduced columns cost +0.2 ms per step in round 2.)
    const int64_t items = ((n_out + M - 1) / M) * (p.kind == 4 ? Co / 64 : Co / (16 * p.nw));
    const bool fused = (p.kind == 2 || (p.kind == 4 && g_conv_gtab)) && split_fused_ok(items, n_out, Co);
    return fused && n_out > 2048 ? (n_out + M - 1) / M : 0;
  }
  return (n_out + M - 1) / M;
}

int scn_conv_plan(int K, int64_t stride, int64_t n_in, int64_t n_out, int Ci, int Co, int out_unique) {
  if (K < 1 || n_out <= 0 || Ci <= 0 || Co <= 0) return 0;
  const ConvPlan p = conv_plan(K, stride, n_in, n_out, Ci, Co, out_unique);
  return 100 * p.kind + p.ksplit;
}

int scn_weight_pack(const float *W, int w_transposed, int w_reverse, int K, int Ci, int Co, void *wpack,
                    void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(K >= 1 && K <= KMAX && Ci > 0 && Co > 0 && wpack != nullptr, "scn_weight_pack: bad arguments");
  if (Ci % 16 != 0 || Co % 16 != 0) return 0;  // the generic kernel reads W itself
  const int64_t total = (int64_t)K * Ci * Co;
  int pg = (int)((total + 255) / 256);
  if (pg > 2048) pg = 2048;
  // fragment order for every kernel but the first synchronous one (which only the > 4 GiB fallback still uses)
  const bool plain = g_conv_variant == 1 || total * 4 >= ((int64_t)1 << 32);
  k_weight_pack<<<pg, 256, 0, stream>>>(W, (float *)wpack, K, Ci, Co, w_transposed, w_reverse, plain ? 0 : 1);
  SCN_CHECK_LAUNCH();
  return 0;
}

int64_t scn_weight_pack_blocks(int K, int Ci, int Co) {
  if (K < 1 || Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0) return 0;  // the generic kernel reads W itself
  return ((int64_t)K * Ci * Co + WPB_ELEMS - 1) / WPB_ELEMS;
}

int scn_weight_pack_flags(int w_transposed, int w_reverse, int K, int Ci, int Co) {
  const bool plain = g_conv_variant == 1 || (int64_t)K * Ci * Co * 4 >= ((int64_t)1 << 32);
  return (w_transposed ? 1 : 0) | (w_reverse ? 2 : 0) | (plain ? 4 : 0);
}

int scn_weight_pack_batch(const void *descs, int n, int64_t total_blocks, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  static_assert(sizeof(WPackDesc) == 40, "scn_wpack_desc layout");
  SCN_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "scn_weight_pack_batch: bad arguments");
  if (n == 0 || total_blocks == 0) return 0;
  SCN_REQUIRE(descs != nullptr, "scn_weight_pack_batch: descriptor table missing");
  k_weight_pack_multi<<<(unsigned)total_blocks, 256, 0, stream>>>((const WPackDesc *)descs, n);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_conv_forward(const float *in, float *out, const float *W, int w_transposed, int w_reverse,
                     const float *bias, const float *residual, float *colstats_out, const scn_bn_bwd_stats *bn_bwd,
                     const int32_t *rules, const int32_t *bases,
                     const int32_t *tile_passes, const int32_t *counts, int K,
                     int64_t stride, int in_col, int out_col, int out_unique, int64_t n_in, int64_t n_out, int Ci,
                     int Co, int64_t ld_in, int64_t ld_out, void *wpack, int wpack_ready, void *workspace,
                     size_t workspace_bytes, void *sk_state, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE((ld_in == 0 || (ld_in >= Ci && ld_in % 4 == 0 && ld_in < (1 << 20))) &&
              (ld_out == 0 || (ld_out >= Co && ld_out % 4 == 0 && ld_out < (1 << 20))), "scn_conv_forward: bad leading dimensions");
  if (ld_in == Ci) ld_in = 0;
  if (ld_out == Co) ld_out = 0;
  StatsArg colstats{colstats_out, nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, (int)ld_in, (int)ld_out};
  if (bn_bwd != nullptr && colstats_out != nullptr) {
    SCN_REQUIRE(bn_bwd->x != nullptr && bn_bwd->save_mean != nullptr && bn_bwd->save_invstd != nullptr,
                "scn_conv_forward: bn_bwd needs x, save_mean and save_invstd");
    colstats = StatsArg{colstats_out, bn_bwd->x, bn_bwd->save_mean, bn_bwd->save_invstd, bn_bwd->weight, bn_bwd->bias,
                        (float)bn_bwd->leakiness, (int)ld_in, (int)ld_out};
  }
  SCN_REQUIRE(K >= 1 && K <= KMAX, "scn_conv_forward: filter volume must be 1..27");
  SCN_REQUIRE((in_col == 0 && out_col == 1) || (in_col == 1 && out_col == 0), "scn_conv_forward: in_col/out_col must be 0/1 or 1/0");
  SCN_REQUIRE(Ci > 0 && Co > 0, "scn_conv_forward: bad plane counts");
  if (n_out <= 0) return 0;
  const ConvPlan p = conv_plan(K, stride, ld_in ? (n_in * ld_in + Ci - 1) / Ci : n_in, n_out, Ci, Co, out_unique && counts != nullptr);
  // column slices of wider tensors: `in` for the pipelined kernels, `out` for the one-contribution-per-row kernel only
  SCN_REQUIRE(ld_in == 0 || p.kind == 2 || p.kind == 4 || p.kind == 5, "scn_conv_forward: ld_in needs planes that are multiples of 16 and < 4 GiB of input");
  SCN_REQUIRE(ld_out == 0 || p.kind == 5, "scn_conv_forward: ld_out is supported by out_unique launches with planes % 64 == 0 only (scn_conv_plan = 5xx)");
  if (p.kind != 0) {
    SCN_REQUIRE(wpack != nullptr, "scn_conv_forward: wpack workspace required");
    float *Wp = (float *)wpack;
    if (!wpack_ready) {
      const int st = scn_weight_pack(W, w_transposed, w_reverse, K, Ci, Co, wpack, stream_);
      if (st) return st;
    }
    if (p.kind == 1) {
      if (p.huge) {  // say so once: this launch runs on the slow first kernel (~45 TFLOP/s instead of ~90)
        static bool warned = false;
        if (!warned) {
          warned = true;
          fprintf(stderr, "libscn_hip: warning: gather-mm input of %lld rows x %d planes (or its weights) exceeds the 4 GiB "
                          "reach of 32-bit buffer offsets; using the synchronous fallback kernel (about half the "
                          "throughput).  Split the batch to stay below 4 GiB per feature tensor.\n", (long long)n_in, Ci);
        }
      }
      switch (p.nw) {
        case 4: return launch_tile<4, 1, 64, 64>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, Ci, Co, stream);
        case 3: return launch_tile<3, 1, 64, 64>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, Ci, Co, stream);
        case 2: return launch_tile<2, 1, 64, 64>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, Ci, Co, stream);
        default: return launch_tile<1, 1, 64, 64>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, Ci, Co, stream);
      }
    }
    if (p.kind == 5) {
      const int ncb = Co / 64;
      const int64_t nblk = expand_waves(n_out, K);  // every output row in exactly one pair: sum_k ceil(count_k / PB) <= this
      const unsigned in_bytes = (unsigned)((n_in - 1) * (colstats.ldi ? colstats.ldi : Ci) * 4 + Ci * 4), wp_bytes = (unsigned)((int64_t)K * Ci * Co * 4);
      // (the plane blocks of a pair block on one XCD: the grid padded to whole groups of eight pair blocks)
      const bool xcd = g_conv_expand_xcd && (ncb > 1 || (K > 1 && K <= 8 && g_conv_expand_xcd > 1)) && nblk < ((int64_t)1 << 27);
      const int64_t grid_e = xcd ? (nblk + 7) / 8 * 8 * ncb : nblk * ncb;
#define SCN_LAUNCH_EXPAND(NB_)                                                                                              \
  k_conv_expand<NB_><<<(unsigned)grid_e, 64, 0, stream>>>(in, out, Wp, bias, residual, colstats, (const int2 *)rules, counts, K, \
                                                               stride, in_col, Ci, Co, ncb, in_bytes, wp_bytes, xcd ? (int)nblk : 0)
#ifdef SCN_TUNING
      if (g_conv_expand_persist && expand_nb() == 1 && nblk * ncb < ((int64_t)1 << 30)) {
        // persistent waves (two per SIMD: 237 registers), pipelined across blocks
        const int64_t items = nblk * ncb;
        const int64_t slots = (int64_t)device_cus() * 8;
        k_conv_expand_p<<<(unsigned)(items < slots ? items : slots), 64, 0, stream>>>(in, out, Wp, bias, residual, colstats, (const int2 *)rules, counts, K,
                                                                                      stride, in_col, Ci, Co, ncb, in_bytes, wp_bytes, (int)items);
      } else
#endif
      switch (expand_nb()) {
        case 4: SCN_LAUNCH_EXPAND(4); break;
        case 2: SCN_LAUNCH_EXPAND(2); break;
        default: SCN_LAUNCH_EXPAND(1);
      }
#undef SCN_LAUNCH_EXPAND
      SCN_CHECK_LAUNCH();
      return 0;
    }
    if (p.ksplit > 1)
      SCN_REQUIRE(workspace != nullptr && workspace_bytes >= (p.kind == 4 ? tile4_split_workspace_bytes(K, p.ksplit, n_out, Ci, Co)
                                                                         : (size_t)p.ksplit * (size_t)n_out * (size_t)Co * 4),
                  "scn_conv_forward: workspace smaller than scn_conv_workspace_bytes()");
    SCN_REQUIRE(p.ksplit == 1 || colstats_out == nullptr || sk_state != nullptr,
                "scn_conv_forward: a launch that splits its offsets emits column statistics only with sk_state");
    float *part = (float *)workspace;
    if (p.kind == 4 && p.ksplit == 1 && workspace != nullptr)
      SCN_REQUIRE(workspace_bytes >= tile4_workspace_bytes(K, n_out, Ci, Co),
                  "scn_conv_forward: workspace smaller than scn_conv_workspace_bytes()");
    if (p.kind == 4)
      return launch_tile4(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, p.ksplit, part,
                          p.ksplit == 1 ? tile_passes : nullptr, (unsigned *)sk_state, stream);
    switch (p.nw) {
      case 4: return launch_tile2_ck<4>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, p.ksplit, part, stream, (unsigned *)sk_state);
      case 3: return launch_tile2_ck<3>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, p.ksplit, part, stream, (unsigned *)sk_state);
      case 2: return launch_tile2_ck<2>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, p.ksplit, part, stream, (unsigned *)sk_state);
      default: return launch_tile2_ck<1>(in, out, Wp, bias, residual, colstats, rules, bases, K, stride, in_col, n_out, n_in, Ci, Co, p.ksplit, part, stream, (unsigned *)sk_state);
    }
  }
  const int n_gran = (int)((n_out + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
  dim3 grid((unsigned)(n_gran < 2048 ? n_gran : 2048), (unsigned)((Co + 63) / 64));  // persistent: weights staged once
  if (g_conv_stem_mfma && Ci <= 4 && Co % 16 == 0 && !w_transposed && !w_reverse && residual == nullptr && colstats.out == nullptr) {
    const int2 *r2 = (const int2 *)rules;
    switch (Ci) {  // the stem as a GEMM on the matrix cores (k_conv_stem)
      case 1: k_conv_stem<1><<<grid, 256, 0, stream>>>(in, out, W, bias, r2, bases, K, stride, in_col, n_out, n_gran, Co); break;
      case 2: k_conv_stem<2><<<grid, 256, 0, stream>>>(in, out, W, bias, r2, bases, K, stride, in_col, n_out, n_gran, Co); break;
      case 3: k_conv_stem<3><<<grid, 256, 0, stream>>>(in, out, W, bias, r2, bases, K, stride, in_col, n_out, n_gran, Co); break;
      default: k_conv_stem<4><<<grid, 256, 0, stream>>>(in, out, W, bias, r2, bases, K, stride, in_col, n_out, n_gran, Co);
    }
    SCN_CHECK_LAUNCH();
    return 0;
  }
  const size_t wbytes = (size_t)K * Ci * 64 * 4;
  if (wbytes <= 40 * 1024)
    k_conv_generic<true><<<grid, 256, wbytes, stream>>>(in, out, W, bias, residual, colstats, (const int2 *)rules, bases, K, stride,
                                                        in_col, n_out, n_gran, Ci, Co, w_transposed, w_reverse);
  else
    k_conv_generic<false><<<grid, 256, 0, stream>>>(in, out, W, bias, residual, colstats, (const int2 *)rules, bases, K, stride,
                                                    in_col, n_out, n_gran, Ci, Co, w_transposed, w_reverse);
  SCN_CHECK_LAUNCH();
  return 0;
}

// slots of the deterministic workspace = workgroups along the pair dimension of the kernel scn_conv_dweight picks
// pairs of one list per k_dw_tile workgroup: enough workgroups per (k, plane block) to fill the chip, few enough that the
// float atomics of the partial sums (one CIB x COB tile per workgroup) stay far below the ~1.3 TB/s atomic rate.  The
// longest list (the centre offset: one pair per site) gets ~2400 pairs per workgroup, an average one about half of that.
// (Every list is cut into the same NUMBER of slices, whatever its length.)  One list only (NiN: the identity list): the slices
// of 26 other lists are missing, so the slices get shorter until ~2048 workgroups have work.
static int dw_pairs_per_wg(int K, int64_t stride, int Ci, int Co) {
  int per = (int)(((stride + g_dw_slices - 1) / g_dw_slices + 63) / 64 * 64);
  if (per < 2048) per = 2048;
  if (per > 16384) per = 16384;
  if (K == 1) {
    const int64_t cb = (Ci % 64 == 0 && Co % 64 == 0) ? (int64_t)(Ci / 64) * (Co / 64) : (int64_t)(Ci / 32) * (Co / 32);
    int64_t p2 = (stride * cb / 2048 + 63) / 64 * 64;
    if (p2 < 512) p2 = 512;
    if (p2 < per) per = (int)p2;
    // (at most 1024 slices: what the in-kernel fixed-order reduction's two levels of 32 cover)
    if ((stride + per - 1) / per > 1024) per = (int)(((stride + 1023) / 1024 + 63) / 64 * 64);
  } else if (K <= 8 && g_dw_k8) {
    // the lists of a strided rulebook PARTITION the fine sites (each ~ stride / K pairs): cut like a submanifold rulebook, whose
    // longest list has `stride` pairs, their workgroups got ~300 pairs = 5 steps between a prologue and 4,096 atomics (the 16
    // launches of this kind in a train step ran at 0.55 of the peak).  ~2,300 workgroups of >= 512 pairs instead.
    const int64_t cb = (Ci % 64 == 0 && Co % 64 == 0) ? (int64_t)(Ci / 64) * (Co / 64) : (int64_t)(Ci / 32) * (Co / 32);
    int64_t gx = (2304 + K * cb - 1) / (K * cb);
    const int64_t most = stride / K / 512, least = (768 + K * cb - 1) / (K * cb);  // (>= 512 pairs each, but one round of the chip at least)
    if (gx > most) gx = most;
    if (gx < least) gx = least;
    const int64_t p2 = ((stride + gx - 1) / gx + 63) / 64 * 64;
    if (p2 > per) per = (int)(p2 < (1 << 24) ? p2 : (1 << 24));  // (never MORE slices than before)
  }
  return per;
}

static int64_t dw_slots(int K, int64_t stride, int64_t n_dout, int Ci, int Co, bool has_bases, int64_t n_in) {
  const bool huge = n_in * Ci * 4 >= ((int64_t)1 << 32) || n_dout * Co * 4 >= ((int64_t)1 << 32) || n_in >= (1 << 24) ||
                    n_dout >= (1 << 24) || stride >= (1 << 28);
  if (Ci % 32 == 0 && Co % 32 == 0 && !huge) {
    const int per = dw_pairs_per_wg(K, stride, Ci, Co);
    return (stride + per - 1) / per;
  }
  if (Ci >= 1 && Ci <= 4 && has_bases && n_dout > 0) {
    const int64_t n_gran = (n_dout + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE;
    return n_gran < 1024 ? n_gran : 1024;
  }
  if (Ci >= 1 && Ci <= 4) return 4 * ((stride + 2047) / 2048);
  return (stride + 255) / 256;
}

size_t scn_conv_dweight_workspace_bytes(int K, int64_t stride, int64_t n_in, int64_t n_dout, int Ci, int Co, int has_bases) {
  if (K < 1 || stride <= 0 || Ci <= 0 || Co <= 0) return 0;
  return (size_t)dw_slots(K, stride, n_dout, Ci, Co, has_bases != 0, n_in) * (size_t)K * Ci * Co * 4;
}

int scn_conv_dweight(const float *in, const float *dout, float *dW, const int32_t *rules,
                     const int32_t *counts, const int32_t *bases, int K, int64_t stride, int in_col, int out_col,
                     int64_t n_in, int64_t n_dout, int Ci, int Co, int64_t ld_in, int64_t ld_dout, void *workspace,
                     size_t workspace_bytes, void *dw_state, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  if (ld_in == 0) ld_in = Ci;
  if (ld_dout == 0) ld_dout = Co;
  SCN_REQUIRE((ld_in == Ci || (ld_in > Ci && ld_in % 4 == 0 && ld_in < (1 << 20))) &&
              (ld_dout == Co || (ld_dout > Co && ld_dout % 4 == 0 && ld_dout < (1 << 20))), "scn_conv_dweight: bad leading dimensions");
  SCN_REQUIRE(K >= 1 && K <= KMAX, "scn_conv_dweight: filter volume must be 1..27");
  SCN_REQUIRE((in_col == 0 && out_col == 1) || (in_col == 1 && out_col == 0), "scn_conv_dweight: in_col/out_col must be 0/1 or 1/0");
  if (stride <= 0) return 0;
  // deterministic mode: per-workgroup partial sums in the (zeroed) workspace, added to dW in a fixed order afterwards
  const int64_t elems = (int64_t)K * Ci * Co;
  const int64_t nslots = workspace ? dw_slots(K, stride, n_dout, Ci, Co, bases != nullptr, n_in) : 0;
  const bool huge_ = n_in * ld_in * 4 >= ((int64_t)1 << 32) || n_dout * ld_dout * 4 >= ((int64_t)1 << 32) || n_in >= (1 << 24) ||
                     n_dout >= (1 << 24) || stride >= (1 << 28);
  // the fixed-order reduction INSIDE k_dw_tile (dw_state: its tickets): no memset, no second launch
  const bool big = g_dw_big && K > 1 && Ci % 128 == 0 && Co % (g_dw_big == 2 ? 64 : 128) == 0 && !huge_ && g_dw_dbg == 0 && !g_dw_direct;
  bool fused = false;
  int grp = DW_GROUP;
  if (workspace && dw_state && g_dw_fused && Ci % 32 == 0 && Co % 32 == 0 && !huge_ && !(g_dw_direct && Ci % 64 == 0 && Co % 64 == 0)) {
    const int per = dw_pairs_per_wg(K, stride, Ci, Co);
    const int64_t gx = (stride + per - 1) / per;
    const int64_t nt = big ? (int64_t)(Ci / 128) * (Co / (g_dw_big == 2 ? 64 : 128))
                           : (Ci % 64 == 0 && Co % 64 == 0) ? (int64_t)(Ci / 64) * (Co / 64) : (int64_t)(Ci / 32) * (Co / 32);
    grp = gx <= DW_GROUP * DW_GROUP ? DW_GROUP : 2 * DW_GROUP;
    fused = gx <= (int64_t)grp * grp && (int64_t)K * nt * (grp + 1) * 4 <= (int64_t)SCN_DW_STATE_BYTES;
  }
  if (workspace) {
    SCN_REQUIRE(workspace_bytes >= (size_t)nslots * (size_t)elems * 4, "scn_conv_dweight: workspace smaller than scn_conv_dweight_workspace_bytes()");
    if (!fused) SCN_CHECK_HIP(hipMemsetAsync(workspace, 0, (size_t)nslots * (size_t)elems * 4, stream));
  }
  const DwOut dwo{dW, (float *)workspace, elems, (int)ld_in, (int)ld_dout, fused ? (unsigned *)dw_state : nullptr, grp};
  auto finish = [&]() -> int {
    if (workspace && !fused) {
      if (elems % 4 == 0 && nslots >= 128 && elems / 4 <= 65536) {
        k_dw_reduce_wide<<<(unsigned)((elems / 4 + 3) / 4), 256, 0, stream>>>((const float *)workspace, dW, elems / 4, (int)nslots);
      } else if (elems % 4 == 0) {
        int64_t g = (elems / 4 + 255) / 256;
        if (g > 4096) g = 4096;
        k_dw_reduce<<<(unsigned)g, 256, 0, stream>>>((const float *)workspace, dW, elems / 4, (int)nslots);
      } else {
        k_dw_reduce1<<<(unsigned)((elems + 255) / 256), 256, 0, stream>>>((const float *)workspace, dW, elems, (int)nslots);
      }
      SCN_CHECK_LAUNCH();
    }
    return 0;
  };
  // (k_dw_tile: 32-bit byte offsets, row numbers through v_mad_u32_u24, 31-bit byte offsets into one rule list)
  const bool huge = n_in * ld_in * 4 >= ((int64_t)1 << 32) || n_dout * ld_dout * 4 >= ((int64_t)1 << 32) || n_in >= (1 << 24) ||
                    n_dout >= (1 << 24) || stride >= (1 << 28);
  SCN_REQUIRE((ld_in == Ci && ld_dout == Co) || (Ci % 32 == 0 && Co % 32 == 0 && !huge),
              "scn_conv_dweight: column slices (ld_in / ld_dout) need planes that are multiples of 32 and < 4 GiB operands");
  if (Ci % 32 == 0 && Co % 32 == 0 && !huge) {
    const unsigned in_bytes = (unsigned)((n_in - 1) * ld_in * 4 + Ci * 4), do_bytes = (unsigned)((n_dout - 1) * ld_dout * 4 + Co * 4);
    const int per = dw_pairs_per_wg(K, stride, Ci, Co);  // slices per list
    const int gx = (int)((stride + per - 1) / per);
#ifdef SCN_TUNING
    // 128 x 128 blocks (tuning key 25): 64 bytes of operand per MFMA from L2 instead of 128 -- and 192-244 registers, two waves
    // per SIMD: 0.938 against 0.883 ms at 128 -> 128 on level 1, 0.208 against 0.144 at level 3 (round 4, tools/ab_dw_det.py
    // with DW_BIG=1).  Key 25 = 2: 128 x 64 blocks (96 bytes per MFMA, 116-140 registers): 3-5.5 % faster than 64 x 64 with float
    // atomics on the big shapes with 128 input planes, -2..+2 % with the fixed-order slots, slower on the coarse levels.  Neither
    // is the product's kernel.
    if (big && g_dw_big == 2) {  // 128 x 64 blocks: 96 bytes of operand per MFMA
      const unsigned nblk = (unsigned)(gx * K * (Ci / 128) * (Co / 64));
      if (fused)
        k_dw_tile<4, 2, 32, 0, 1><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
      else
        k_dw_tile<4, 2, 32><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
    } else if (big) {
      const unsigned nblk = (unsigned)(gx * K * (Ci / 128) * (Co / 128));
      if (fused)
        k_dw_tile<4, 4, 32, 0, 1><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
      else
        k_dw_tile<4, 4, 32><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
    } else
#endif
    if (Ci % 64 == 0 && Co % 64 == 0) {
      const unsigned nblk = (unsigned)(gx * K * (Ci / 64) * (Co / 64));
      if (g_dw_direct && g_dw_dbg == 0) {
        if (in_col)
          k_dw_direct<1><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, Ci, Co, gx, in_bytes, do_bytes, K);
        else
          k_dw_direct<0><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, Ci, Co, gx, in_bytes, do_bytes, K);
        SCN_CHECK_LAUNCH();
        return finish();
      }
#ifdef SCN_TUNING
      if (g_dw_dbg == 100)  // A/B: 32 pairs per step
        k_dw_tile<2, 2, 32><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
      else if (g_dw_dbg == 103)  // timing only: every row load reads row 0 (no gather traffic)
        k_dw_tile<2, 2, 64, 1><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
      else if (g_dw_dbg == 101)
        k_dw_tile<2, 2, 64><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col + 2, Ci, Co, gx, in_bytes, do_bytes, -K);
      else
#endif
      if (fused)
        k_dw_tile<2, 2, 64, 0, 1><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
      else
        k_dw_tile<2, 2, 64><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
    } else {
      const unsigned nblk = (unsigned)(gx * K * (Ci / 32) * (Co / 32));
      if (fused)
        k_dw_tile<1, 1, 64, 0, 1><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
      else
        k_dw_tile<1, 1, 64><<<nblk, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, gx, in_bytes, do_bytes, K);
    }
    SCN_CHECK_LAUNCH();
    return finish();
  }
  if (Ci >= 1 && Ci <= 4 && bases != nullptr && n_dout > 0) {
    const int n_gran = (int)((n_dout + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
    dim3 grid((unsigned)(n_gran < 1024 ? n_gran : 1024), (unsigned)((Co + 63) / 64));
    if (g_conv_stem_mfma && Co % 16 == 0) {  // on the matrix cores (k_dw_stem)
      const int2 *r2 = (const int2 *)rules;
      switch (Ci) {
        case 1: k_dw_stem<1><<<grid, 256, 0, stream>>>(in, dout, dwo, r2, bases, K, stride, in_col, n_gran, n_dout, Co); break;
        case 2: k_dw_stem<2><<<grid, 256, 0, stream>>>(in, dout, dwo, r2, bases, K, stride, in_col, n_gran, n_dout, Co); break;
        case 3: k_dw_stem<3><<<grid, 256, 0, stream>>>(in, dout, dwo, r2, bases, K, stride, in_col, n_gran, n_dout, Co); break;
        default: k_dw_stem<4><<<grid, 256, 0, stream>>>(in, dout, dwo, r2, bases, K, stride, in_col, n_gran, n_dout, Co);
      }
      SCN_CHECK_LAUNCH();
      return finish();
    }
    switch (Ci) {
      case 1: k_dw_rows<1><<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, bases, K, stride, in_col, n_gran, n_dout, Co); break;
      case 2: k_dw_rows<2><<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, bases, K, stride, in_col, n_gran, n_dout, Co); break;
      case 3: k_dw_rows<3><<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, bases, K, stride, in_col, n_gran, n_dout, Co); break;
      default: k_dw_rows<4><<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, bases, K, stride, in_col, n_gran, n_dout, Co);
    }
    SCN_CHECK_LAUNCH();
    return finish();
  }
  if (Ci >= 1 && Ci <= 4) {
    const int chunk = 2048;
    dim3 grid((unsigned)((stride + chunk - 1) / chunk), K);
    switch (Ci) {
      case 1: k_dw_narrow<1><<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Co, chunk); break;
      case 2: k_dw_narrow<2><<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Co, chunk); break;
      case 3: k_dw_narrow<3><<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Co, chunk); break;
      default: k_dw_narrow<4><<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Co, chunk);
    }
    SCN_CHECK_LAUNCH();
    return finish();
  }
  const int chunk = 256;
  dim3 grid((unsigned)((stride + chunk - 1) / chunk), K);
  k_dw_generic<<<grid, 256, 0, stream>>>(in, dout, dwo, (const int2 *)rules, counts, stride, in_col, Ci, Co, chunk);
  SCN_CHECK_LAUNCH();
  return finish();
}

}  // extern "C"
