// Dense [rows, planes] kernels of the hot path: batch norm over active sites,
// input/output layers, column sums, weight transposition (gfx950).
//
// All of these are HBM-bound streaming passes: rows are walked with 16-byte
// loads along the plane dimension (coalesced), reductions over rows are
// per-thread registers -> LDS -> per-block partial -> fp64 finalize, so every
// CU takes part (the reference's BN kernels run on at most 16 blocks,
// sparseconvnet/SCN/CUDA/BatchNormalization.cu:110-114).
#include "scn_common.h"

namespace scn {

constexpr int BN_MAX_BLOCKS = 1024;

// Reduce column sums of two per-row quantities over a block's row range.
// Thread (cg, slot): column group cg (VEC planes), row slot; rows strided by nslots.
template <int VEC, class F>
__device__ __forceinline__ void bn_reduce_rows(int64_t N, int C, float *partial, F &&load2) {
  // partial layout: [gridDim.x][2][C]
  __shared__ float red[2][256 * 4];
  const int cg_total = C / VEC;
  const int cgb = min(cg_total, 256);
  const int nslots = 256 / cgb;
  const int cg = blockIdx.y * cgb + (threadIdx.x % cgb);
  const int slot = threadIdx.x / cgb;
  const bool active = slot < nslots && cg < cg_total;
  float a[VEC], b[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) a[v] = b[v] = 0.f;
  const int64_t rows_per_blk = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(N, r0 + rows_per_blk);
  if (active)
    for (int64_t r = r0 + slot; r < r1; r += nslots) load2(r, cg * VEC, a, b);
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    red[0][threadIdx.x * VEC + v] = a[v];
    red[1][threadIdx.x * VEC + v] = b[v];
  }
  __syncthreads();
  if (slot == 0 && cg < cg_total) {
    for (int s = 1; s < nslots; ++s)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        a[v] += red[0][(s * cgb + threadIdx.x) * VEC + v];
        b[v] += red[1][(s * cgb + threadIdx.x) * VEC + v];
      }
    float *p = partial + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      p[cg * VEC + v] = a[v];
      p[C + cg * VEC + v] = b[v];
    }
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) k_bn_stats(const float *x, int64_t N, int C, float *partial) {
  bn_reduce_rows<VEC>(N, C, partial, [&](int64_t r, int c, float *a, float *b) {
    const float *px = x + r * C + c;
    float v[VEC];
    if (VEC == 4) {
      const float4 t = *(const float4 *)px;
      v[0] = t.x; v[1 % VEC] = t.y; v[2 % VEC] = t.z; v[3 % VEC] = t.w;
    } else {
      v[0] = px[0];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      a[i] += v[i];
      b[i] += v[i] * v[i];
    }
  });
}

// Sum the per-block partials of 16 columns with 256 threads (16 columns x 16 partial lanes), in fp64.
// Returns the two column sums to the threads with bl == 0.
// COHERENT: the partials were written by other workgroups of THIS launch (write-through stores; k_bn_fold_finalize): read them
// with agent-scope loads -- the L2 caches of two XCDs are not coherent.  cgroup: which 16 columns (blockIdx.x in the plain kernels).
template <bool COHERENT = false>
__device__ __forceinline__ bool bn_sum_partials(const float *partial, int nblk, int C, int &c, double &s0, double &s1, int cgroup) {
  __shared__ double red[2][16][17];
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  c = cgroup * 16 + cl;
  double a = 0, b = 0;
  auto ld = [](const float *q) __attribute__((always_inline)) {
    return COHERENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
  };
  if (c < C) {
    // up to 64 dependent fp64 adds per thread: the LOADS are independent, so eight iterations' worth are requested before
    // the first add (same summation order; the finalize kernels took 26 us of pure L2 latency per launch before)
    const float *p0 = partial + c;
    int blk = bl;
    for (; blk + 7 * 16 < nblk; blk += 8 * 16) {
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        va[u] = ld(p0 + (int64_t)(blk + 16 * u) * 2 * C);
        vb[u] = ld(p0 + (int64_t)(blk + 16 * u) * 2 * C + C);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += (double)va[u];
        b += (double)vb[u];
      }
    }
    for (; blk < nblk; blk += 16) {
      a += (double)ld(p0 + (int64_t)blk * 2 * C);
      b += (double)ld(p0 + (int64_t)blk * 2 * C + C);
    }
  }
  red[0][bl][cl] = a;
  red[1][bl][cl] = b;
  __syncthreads();
  if (bl != 0 || c >= C) return false;
  s0 = 0; s1 = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    s0 += red[0][j][cl];
    s1 += red[1][j][cl];
  }
  return true;
}

// arithmetic of sparseconvnet/SCN/CPU/BatchNormalization.cpp:19-52 with the row sums finished in fp64.
// The partial sums [nblk][2][Cp] may cover a SEGMENT of the planes only (col0 .. col0 + Cp): statistics handed over by
// the kernel that produced those planes (scn_conv_forward's colstats), one finalize launch per segment.
__device__ __forceinline__ void bn_finalize_col(int c, double s, double ss, int64_t N, int C, float *save_mean, float *save_invstd,
                                                float *running_mean, float *running_var, const float *weight, const float *bias,
                                                float eps, float momentum, int train, float *wb);
template <bool COHERENT>
__device__ __forceinline__ void bn_finalize_group(const float *partial, int nblk, int64_t N, int Cp, int col0, int C, float *save_mean,
                                                  float *save_invstd, float *running_mean, float *running_var,
                                                  const float *weight, const float *bias, float eps, float momentum,
                                                  int train, float *wb, int cgroup) {
  int cl;
  double s = 0, ss = 0;
  if (!bn_sum_partials<COHERENT>(partial, train ? nblk : 0, Cp, cl, s, ss, cgroup)) return;
  bn_finalize_col(col0 + cl, s, ss, N, C, save_mean, save_invstd, running_mean, running_var, weight, bias, eps, momentum, train, wb);
}
__device__ __forceinline__ void bn_finalize_col(int c, double s, double ss, int64_t N, int C, float *save_mean, float *save_invstd,
                                                float *running_mean, float *running_var, const float *weight, const float *bias,
                                                float eps, float momentum, int train, float *wb) {
  float mean, invstd;
  if (train) {
    const double m = s / (double)N;
    const double var_n = ss - m * m * (double)N;  // "saveInvStd" before the rsqrt: sum of squared deviations
    mean = (float)m;
    running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * mean;
    running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)(var_n / (double)(N - 1));
    invstd = (float)(1.0 / sqrt(var_n / (double)N + (double)eps));
  } else {
    mean = running_mean[c];
    invstd = powf(running_var[c] + eps, -0.5f);
  }
  save_mean[c] = mean;
  save_invstd[c] = invstd;
  // y = fma(x, w, b) with b = fma(-mean, w, beta): written with explicit fma here, in k_bn_apply and in the backward
  // kernels, so that the backward pass can recompute the sign of y from x bit for bit (SCN_BN_MASK_FROM_X)
  const float w = invstd * (weight ? weight[c] : 1.f);
  wb[c] = w;
  wb[C + c] = fmaf(-mean, w, bias ? bias[c] : 0.f);
}

__global__ void __launch_bounds__(256)
    k_bn_finalize(const float *partial, int nblk, int64_t N, int Cp, int col0, int C, float *save_mean,
                  float *save_invstd, float *running_mean, float *running_var,
                  const float *weight, const float *bias, float eps, float momentum,
                  int train, float *wb) {
  bn_finalize_group<false>(partial, nblk, N, Cp, col0, C, save_mean, save_invstd, running_mean, running_var, weight, bias, eps, momentum,
                           train, wb, blockIdx.x);
}

template <int VEC>
__global__ void __launch_bounds__(256)
    k_bn_apply(const float *x, float *y, int64_t total_vec, int C, const float *wb, float leak) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i * VEC) % C);
    if (VEC == 4) {
      const float4 v = ((const float4 *)x)[i];
      const float4 w = *(const float4 *)(wb + c);
      const float4 b = *(const float4 *)(wb + C + c);
      float4 o;
      o.x = fmaf(v.x, w.x, b.x); o.y = fmaf(v.y, w.y, b.y); o.z = fmaf(v.z, w.z, b.z); o.w = fmaf(v.w, w.w, b.w);
      o.x = o.x > 0 ? o.x : o.x * leak; o.y = o.y > 0 ? o.y : o.y * leak;
      o.z = o.z > 0 ? o.z : o.z * leak; o.w = o.w > 0 ? o.w : o.w * leak;
      ((float4 *)y)[i] = o;
    } else {
      float o = fmaf(x[i], wb[c], wb[C + c]);
      y[i] = o > 0 ? o : o * leak;
    }
  }
}

// backward pass 1: masked gradient, reduce sum(d) and sum((x-mean) d)   (sparseconvnet/SCN/CPU/BatchNormalization.cpp:75-87)
// FROM_X = false: mask from y as the reference does, masked gradient written to d_masked (d_out itself = the
//                 reference's in-place rewrite, or d_x).  16 B per element.
// FROM_X = true : the sign of y is recomputed from x with the forward pass's own arithmetic (same fma, same w and b:
//                 bit-identical), nothing is written and y is not read.  8 B per element.
template <int VEC, bool FROM_X>
__global__ void __launch_bounds__(256)
    k_bn_bwd_stats(const float *x, const float *y, const float *d_out, float *d_masked, const float *save_mean,
                   const float *save_invstd, const float *weight, const float *bias,
                   int64_t N, int C, float leak, float *partial) {
  bn_reduce_rows<VEC>(N, C, partial, [&](int64_t r, int c, float *a, float *b) {
    const int64_t o = r * C + c;
    float xv[VEC], d[VEC], m[VEC];
    if (VEC == 4) {
      const float4 t = *(const float4 *)(x + o), u = *(const float4 *)(d_out + o), v = *(const float4 *)(save_mean + c);
      xv[0] = t.x; xv[1 % VEC] = t.y; xv[2 % VEC] = t.z; xv[3 % VEC] = t.w;
      d[0] = u.x; d[1 % VEC] = u.y; d[2 % VEC] = u.z; d[3 % VEC] = u.w;
      m[0] = v.x; m[1 % VEC] = v.y; m[2 % VEC] = v.z; m[3 % VEC] = v.w;
    } else {
      xv[0] = x[o]; d[0] = d_out[o]; m[0] = save_mean[c];
    }
    if (FROM_X) {
      if (leak != 1.f) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float w = save_invstd[c + i] * (weight ? weight[c + i] : 1.f);
          const float yy = fmaf(xv[i], w, fmaf(-m[i], w, bias ? bias[c + i] : 0.f));
          d[i] *= yy > 0 ? 1.f : leak;
        }
      }
    } else {
      float yv[VEC];
      if (VEC == 4) {
        const float4 t = *(const float4 *)(y + o);
        yv[0] = t.x; yv[1 % VEC] = t.y; yv[2 % VEC] = t.z; yv[3 % VEC] = t.w;
      } else {
        yv[0] = y[o];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) d[i] *= yv[i] > 0 ? 1.f : leak;
      if (VEC == 4) *(float4 *)(d_masked + o) = make_float4(d[0], d[1 % VEC], d[2 % VEC], d[3 % VEC]);
      else d_masked[o] = d[0];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      a[i] += d[i];
      b[i] += (xv[i] - m[i]) * d[i];
    }
  });
}

// (sparseconvnet/SCN/CPU/BatchNormalization.cpp:88-93,:103-106); coef = [gradMean | k | invstd*gamma | w | b]
__device__ __forceinline__ void bn_bwd_finalize_col(int c, double s, double dp, int64_t N, int C, const float *save_mean,
                                                    const float *save_invstd, const float *weight, const float *bias, float *d_weight,
                                                    float *d_bias, float *coef, int accumulate);
template <bool COHERENT>
__device__ __forceinline__ void bn_bwd_finalize_group(const float *partial, int nblk, int64_t N, int C, const float *save_mean,
                                                      const float *save_invstd, const float *weight, const float *bias, float *d_weight,
                                                      float *d_bias, float *coef, int accumulate, int cgroup) {
  int c;
  double s = 0, dp = 0;
  if (!bn_sum_partials<COHERENT>(partial, nblk, C, c, s, dp, cgroup)) return;
  bn_bwd_finalize_col(c, s, dp, N, C, save_mean, save_invstd, weight, bias, d_weight, d_bias, coef, accumulate);
}
__device__ __forceinline__ void bn_bwd_finalize_col(int c, double s, double dp, int64_t N, int C, const float *save_mean,
                                                    const float *save_invstd, const float *weight, const float *bias, float *d_weight,
                                                    float *d_bias, float *coef, int accumulate) {
  const float invstd = save_invstd[c];
  if (d_bias) d_bias[c] = (accumulate ? d_bias[c] : 0.f) + (float)s;
  if (d_weight) d_weight[c] = (accumulate ? d_weight[c] : 0.f) + (float)(dp * (double)invstd);
  coef[c] = (float)(s / (double)N);
  coef[C + c] = (float)(dp * (double)invstd * (double)invstd / (double)N);
  const float w = invstd * (weight ? weight[c] : 1.f);
  coef[2 * C + c] = w;
  coef[3 * C + c] = w;                                                  // forward's w ...
  coef[4 * C + c] = fmaf(-save_mean[c], w, bias ? bias[c] : 0.f);       // ... and b: y = fma(x, w, b)
}

__global__ void __launch_bounds__(256)
    k_bn_bwd_finalize(const float *partial, int nblk, int64_t N, int C, const float *save_mean, const float *save_invstd,
                      const float *weight, const float *bias, float *d_weight, float *d_bias, float *coef,
                      int accumulate) {
  bn_bwd_finalize_group<false>(partial, nblk, N, C, save_mean, save_invstd, weight, bias, d_weight, d_bias, coef, accumulate, blockIdx.x);
}

template <int VEC, bool FROM_X>
__global__ void __launch_bounds__(256)
    k_bn_bwd_apply(const float *x, const float *d_out, float *d_x, const float *save_mean,
                   int64_t total_vec, int C, const float *coef, float leak, const float *add = nullptr, int ld_add = 0) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i * VEC) % C);
    if (VEC == 4) {
      const float4 xv = ((const float4 *)x)[i];
      float4 d = ((const float4 *)d_out)[i];
      const float4 m = *(const float4 *)(save_mean + c);
      const float4 gm = *(const float4 *)(coef + c);
      const float4 kk = *(const float4 *)(coef + C + c);
      const float4 sc = *(const float4 *)(coef + 2 * C + c);
      if (FROM_X && leak != 1.f) {  // d_out is the raw gradient: apply the (leaky) ReLU mask recomputed from x
        const float4 w = *(const float4 *)(coef + 3 * C + c);
        const float4 b = *(const float4 *)(coef + 4 * C + c);
        d.x *= fmaf(xv.x, w.x, b.x) > 0 ? 1.f : leak; d.y *= fmaf(xv.y, w.y, b.y) > 0 ? 1.f : leak;
        d.z *= fmaf(xv.z, w.z, b.z) > 0 ? 1.f : leak; d.w *= fmaf(xv.w, w.w, b.w) > 0 ? 1.f : leak;
      }
      float4 o;
      o.x = (d.x - gm.x - (xv.x - m.x) * kk.x) * sc.x;
      o.y = (d.y - gm.y - (xv.y - m.y) * kk.y) * sc.y;
      o.z = (d.z - gm.z - (xv.z - m.z) * kk.z) * sc.z;
      o.w = (d.w - gm.w - (xv.w - m.w) * kk.w) * sc.w;
      if (add) {  // the gradient arriving at x along another path (shortcut): one read here instead of an add kernel
        // (ld_add: it may be a column slice of a wider tensor -- the left half of a joined gradient)
        const float4 e = ld_add ? *(const float4 *)(add + ((i * 4) / C) * (int64_t)ld_add + c) : ((const float4 *)add)[i];
        o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
      }
      ((float4 *)d_x)[i] = o;
    } else {
      float d = d_out[i];
      if (FROM_X && leak != 1.f) d *= fmaf(x[i], coef[3 * C + c], coef[4 * C + c]) > 0 ? 1.f : leak;
      d_x[i] = (d - coef[c] - (x[i] - save_mean[c]) * coef[C + c]) * coef[2 * C + c] +
               (add ? (ld_add ? add[(i / C) * (int64_t)ld_add + c] : add[i]) : 0.f);
    }
  }
}

// ------------------------------------------------ batch norm, few rows ---
// The coarse UNet levels (a few hundred to a few thousand sites) spent their batch norms in launch overhead: three
// launches per pass, each a few microseconds for a tensor that fits in L2.  One launch per pass here: a workgroup owns
// 16 planes (4 column groups of 4) x all rows -- 64 row slots -- reduces them (per-thread fp32, fp64 across the 64
// slots), finalizes (same arithmetic as k_bn_finalize / k_bn_bwd_finalize), and applies with the rows coming back from L2.
constexpr int64_t BN_SMALL_ROWS = 2048;  // measured: at 7k rows x 256 planes the 16 workgroups of this kernel stream slower than
                                         // the three full-chip launches (bn 6.0 against 5.0 ms per step with the limit at 16384)

__device__ __forceinline__ void bn_small_reduce(float4 a, float4 b, double (&s0)[4], double (&s1)[4], bool &owner) {
  __shared__ float red[2][64][16];
  const int cgl = threadIdx.x & 3, slot = threadIdx.x >> 2;
  *(float4 *)&red[0][slot][cgl * 4] = a;
  *(float4 *)&red[1][slot][cgl * 4] = b;
  __syncthreads();
  owner = slot == 0;
  if (owner) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0[e] = 0; s1[e] = 0; }
    for (int j = 0; j < 64; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0[e] += (double)red[0][j][cgl * 4 + e];
        s1[e] += (double)red[1][j][cgl * 4 + e];
      }
  }
}

__global__ void __launch_bounds__(256)
    k_bn_small_fwd(const float *__restrict__ x, float *__restrict__ y, int64_t N, int C, float *save_mean,
                   float *save_invstd, float *running_mean, float *running_var, const float *weight,
                   const float *bias, float eps, float momentum, float leak) {
  __shared__ float wb[2][16];
  const int cgl = threadIdx.x & 3, slot = threadIdx.x >> 2;
  const int c0 = blockIdx.x * 16 + cgl * 4;
  const bool live = c0 < C;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (live)
    // (four rows in flight per thread: one dependent L2 round trip per row made a 1,774-row launch 29 us)
#pragma unroll 4
    for (int64_t r = slot; r < N; r += 64) {
      const float4 v = *(const float4 *)(x + r * C + c0);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      b.x += v.x * v.x; b.y += v.y * v.y; b.z += v.z * v.z; b.w += v.w * v.w;
    }
  double s[4], ss[4];
  bool owner;
  bn_small_reduce(a, b, s, ss, owner);
  if (owner && live) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      const double m = s[e] / (double)N;
      const double var_n = ss[e] - m * m * (double)N;
      const float mean = (float)m;
      running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * mean;
      running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)(var_n / (double)(N - 1));
      const float invstd = (float)(1.0 / sqrt(var_n / (double)N + (double)eps));
      save_mean[c] = mean;
      save_invstd[c] = invstd;
      const float w = invstd * (weight ? weight[c] : 1.f);
      wb[0][cgl * 4 + e] = w;
      wb[1][cgl * 4 + e] = fmaf(-mean, w, bias ? bias[c] : 0.f);
    }
  }
  __syncthreads();
  if (!live) return;
  const float4 w = *(const float4 *)&wb[0][cgl * 4], bb = *(const float4 *)&wb[1][cgl * 4];
#pragma unroll 4
  for (int64_t r = slot; r < N; r += 64) {
    const float4 v = *(const float4 *)(x + r * C + c0);
    float4 o;
    o.x = fmaf(v.x, w.x, bb.x); o.y = fmaf(v.y, w.y, bb.y); o.z = fmaf(v.z, w.z, bb.z); o.w = fmaf(v.w, w.w, bb.w);
    o.x = o.x > 0 ? o.x : o.x * leak; o.y = o.y > 0 ? o.y : o.y * leak;
    o.z = o.z > 0 ? o.z : o.z * leak; o.w = o.w > 0 ? o.w : o.w * leak;
    *(float4 *)(y + r * C + c0) = o;
  }
}

// backward, mask recomputed from x (SCN_BN_MASK_FROM_X): d_out is read twice and never written
__global__ void __launch_bounds__(256)
    k_bn_small_bwd(const float *__restrict__ x, const float *__restrict__ d_out, float *__restrict__ d_x, int64_t N, int C,
                   const float *save_mean, const float *save_invstd, const float *weight, const float *bias,
                   float *d_weight, float *d_bias, float leak, int accumulate, const float *__restrict__ add = nullptr,
                   int ld_add = 0) {
  __shared__ float coef[3][16];
  const int cgl = threadIdx.x & 3, slot = threadIdx.x >> 2;
  const int c0 = blockIdx.x * 16 + cgl * 4;
  const bool live = c0 < C;
  float m[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m[e] = save_mean[c0 + e];
      w[e] = save_invstd[c0 + e] * (weight ? weight[c0 + e] : 1.f);
      bb[e] = fmaf(-m[e], w[e], bias ? bias[c0 + e] : 0.f);
    }
  }
  auto masked = [&](const float4 &xv, float4 d) {
    if (leak != 1.f) {
      d.x *= fmaf(xv.x, w[0], bb[0]) > 0 ? 1.f : leak; d.y *= fmaf(xv.y, w[1], bb[1]) > 0 ? 1.f : leak;
      d.z *= fmaf(xv.z, w[2], bb[2]) > 0 ? 1.f : leak; d.w *= fmaf(xv.w, w[3], bb[3]) > 0 ? 1.f : leak;
    }
    return d;
  };
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (live)
    // (four rows in flight per thread: one dependent L2 round trip per row made a 1,774-row launch 29 us)
#pragma unroll 4
    for (int64_t r = slot; r < N; r += 64) {
      const float4 xv = *(const float4 *)(x + r * C + c0);
      const float4 d = masked(xv, *(const float4 *)(d_out + r * C + c0));
      a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
      b.x += (xv.x - m[0]) * d.x; b.y += (xv.y - m[1]) * d.y; b.z += (xv.z - m[2]) * d.z; b.w += (xv.w - m[3]) * d.w;
    }
  double s[4], dp[4];
  bool owner;
  bn_small_reduce(a, b, s, dp, owner);
  if (owner && live) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      const float invstd = save_invstd[c];
      if (d_bias) d_bias[c] = (accumulate ? d_bias[c] : 0.f) + (float)s[e];
      if (d_weight) d_weight[c] = (accumulate ? d_weight[c] : 0.f) + (float)(dp[e] * (double)invstd);
      coef[0][cgl * 4 + e] = (float)(s[e] / (double)N);
      coef[1][cgl * 4 + e] = (float)(dp[e] * (double)invstd * (double)invstd / (double)N);
      coef[2][cgl * 4 + e] = invstd * (weight ? weight[c] : 1.f);
    }
  }
  __syncthreads();
  if (!live) return;
  const float4 gm = *(const float4 *)&coef[0][cgl * 4], kk = *(const float4 *)&coef[1][cgl * 4],
               sc = *(const float4 *)&coef[2][cgl * 4];
#pragma unroll 4
  for (int64_t r = slot; r < N; r += 64) {
    const float4 xv = *(const float4 *)(x + r * C + c0);
    const float4 d = masked(xv, *(const float4 *)(d_out + r * C + c0));
    float4 o;
    o.x = (d.x - gm.x - (xv.x - m[0]) * kk.x) * sc.x;
    o.y = (d.y - gm.y - (xv.y - m[1]) * kk.y) * sc.y;
    o.z = (d.z - gm.z - (xv.z - m[2]) * kk.z) * sc.z;
    o.w = (d.w - gm.w - (xv.w - m[3]) * kk.w) * sc.w;
    if (add) {
      const float4 e = *(const float4 *)(add + r * (int64_t)(ld_add ? ld_add : C) + c0);
      o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
    }
    *(float4 *)(d_x + r * C + c0) = o;
  }
}

// ------------------------------------------------------ input / output ---
// out[s] = sum_j mult * in[order[q0+j]]  (sparseconvnet/SCN/CUDA/IOLayers.cu:17-31)
__global__ void k_input_fwd(const float *feats, float *out, const int32_t *order,
                            const int32_t *site_start, int64_t N, int C, int average) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * C;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / C;
    const int c = (int)(t % C);
    const int q0 = site_start[s], cnt = site_start[s + 1] - q0;
    const float mult = (average && cnt > 0) ? 1.f / (float)cnt : 1.f;
    float acc = 0.f;
    for (int j = 0; j < cnt; ++j) acc += mult * feats[(int64_t)order[q0 + j] * C + c];
    out[t] = acc;
  }
}
// d_in[p] = mult(site) * d_out[site(p)]  (sparseconvnet/SCN/CUDA/IOLayers.cu:46-60, without atomics:
// every input row belongs to exactly one site)
__global__ void k_input_bwd(float *d_feats, const float *d_out, const int32_t *site_of_row,
                            const int32_t *site_start, int64_t P, int C, int average) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < P * C;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / C;
    const int c = (int)(t % C);
    const int s = site_of_row[p];
    float mult = 1.f;
    if (average) {
      const int cnt = site_start[s + 1] - site_start[s];
      mult = cnt > 0 ? 1.f / (float)cnt : 1.f;
    }
    d_feats[t] = mult * d_out[(int64_t)s * C + c];
  }
}
// out[p] = in[site(p)]  (sparseconvnet/SCN/CUDA/IOLayers.cpp:108-129)
template <int VEC>
__global__ void k_output_fwd(const float *in, float *out, const int32_t *site_of_row, int64_t P, int C) {
  const int cv = C / VEC;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < P * cv;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / cv;
    const int c = (int)(t % cv) * VEC;
    const int64_t s = site_of_row[p];
    if (VEC == 4) *(float4 *)(out + p * C + c) = *(const float4 *)(in + s * C + c);
    else out[p * C + c] = in[s * C + c];
  }
}
// d_in[s] = sum_j d_out[order[q0+j]]  (sparseconvnet/SCN/CUDA/IOLayers.cpp:131-154)
template <int VEC>
__global__ void k_output_bwd(float *d_in, const float *d_out, const int32_t *order,
                             const int32_t *site_start, int64_t N, int C) {
  const int cv = C / VEC;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * cv;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / cv;
    const int c = (int)(t % cv) * VEC;
    const int q0 = site_start[s], cnt = site_start[s + 1] - q0;
    if (VEC == 4) {
      float4 acc = make_float4(0, 0, 0, 0);
      for (int j = 0; j < cnt; ++j) {
        const float4 v = *(const float4 *)(d_out + (int64_t)order[q0 + j] * C + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *(float4 *)(d_in + s * C + c) = acc;
    } else {
      float acc = 0.f;
      for (int j = 0; j < cnt; ++j) acc += d_out[(int64_t)order[q0 + j] * C + c];
      d_in[s * C + c] = acc;
    }
  }
}

__global__ void k_column_sum(const float *x, float *out, int64_t rows, int C) {
  // grid.x blocks over row chunks; thread handles column c = threadIdx.x + k*blockDim.x
  const int64_t rows_per_blk = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc += x[r * C + c];
    atomicAdd(out + c, acc);
  }
}

static inline int grid_for(int64_t n, int block = 256, int cap = 4096) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
static inline int bn_blocks(int64_t N) {
  int64_t b = (N + 63) / 64;  // >= 64 rows per block
  if (b < 1) b = 1;
  if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
  return (int)b;
}

}  // namespace scn
using namespace scn;

extern "C" {

size_t scn_bn_temp_bytes(int C) { return ((size_t)BN_MAX_BLOCKS * 2 * C + 5 * (size_t)C) * 4 + 256; }

// [nblk][2][Cp] -> [NF][2][Cp]: fold f sums the partials f, f + NF, f + 2 NF, ... (fp64, fixed order).  A convolution over
// 600k sites hands over one partial per 128-row tile (4,701 of them; 18,727 per-wave partials from k_conv_expand): the
// 16-lane finalize would walk them serially on C/16 workgroups, which costs more than the statistics pass it replaces.
constexpr int BN_FOLD = 64;
__global__ void __launch_bounds__(256) k_bn_fold(const float *__restrict__ partial, int nblk, int Cp, float *__restrict__ out) {
  const int f = blockIdx.y;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < 2 * Cp; c += gridDim.x * 256) {
    double a = 0;
    int b = f;
    for (; b + 7 * BN_FOLD < nblk; b += 8 * BN_FOLD) {  // eight loads in flight, same summation order (see bn_sum_partials)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * BN_FOLD) * 2 * Cp + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += (double)v[u];
    }
    for (; b < nblk; b += BN_FOLD) a += (double)partial[(int64_t)b * 2 * Cp + c];
    out[(int64_t)f * 2 * Cp + c] = (float)a;
  }
}

// fold AND finalize in one launch (round 5: the fold and the finalize were two launches of 5-7 us each in front of every batch-norm
// apply pass whose statistics come from a convolution over >= 33 k rows -- 62 launch pairs per train step).  Workgroup (bx, f) folds
// the columns bx * 128 .. + 127 of fold f exactly as k_bn_fold does, stores them write-through and draws the ticket of its column
// range (words of the stream's control block, zero before and after); the LAST of a range's 64 workgroups to arrive finalizes those
// 128 columns: each thread requests the 64 folded partials of its sum at once (agent-scope loads: the L2s of two XCDs are not
// coherent) and adds them in bn_sum_partials' order -- bit-identical to k_bn_fold + k_bn_finalize / k_bn_bwd_finalize.  (The first
// version let ONE workgroup finalize all column groups one after the other: 16 dependent round trips for 256 planes, 0.7 ms per
// train step SLOWER than the two launches; this one: -0.17 ms, tools/ab_step.py BN_FOLD_FUSED=0,1: 44.82 / 44.86 -> 44.65 / 44.67.)
extern "C++" {
struct BnFinArgs {
  int64_t N;
  int Cp, col0, C, accumulate;
  float *save_mean, *save_invstd, *running_mean, *running_var;
  const float *weight, *bias;
  float *d_weight, *d_bias, *out;  // out: wb (forward) / coef (backward)
  float eps, momentum;
};
template <bool BWD>
__global__ void __launch_bounds__(256) k_bn_fold_finalize(const float *__restrict__ partial, int nblk, float *__restrict__ folded,
                                                          unsigned *__restrict__ tickets, BnFinArgs a) {
  // workgroup (bx, f): columns bx * 128 .. + 127 -- thread t: column t & 127, its sum (t < 128) or its second sum (t >= 128) -- of
  // fold f.  The 64 folds of a column range share a ticket; the last of them to arrive finalizes those 128 columns.
  __shared__ int last_s;
  __shared__ double sums[2][128];
  const int f = blockIdx.y, Cp = a.Cp, cl = threadIdx.x & 127, half = threadIdx.x >> 7;
  const int c = blockIdx.x * 128 + cl;
  const bool live = c < Cp;
  const int64_t idx = (int64_t)half * Cp + c;
  if (live) {
    double acc = 0;
    int b = f;
    for (; b + 7 * BN_FOLD < nblk; b += 8 * BN_FOLD) {  // (k_bn_fold's order)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * BN_FOLD) * 2 * Cp + idx];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; b < nblk; b += BN_FOLD) acc += (double)partial[(int64_t)b * 2 * Cp + idx];
    __hip_atomic_store(folded + (int64_t)f * 2 * Cp + idx, (float)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = t == (unsigned)BN_FOLD - 1u;
    if (last_s) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last_s) return;
  asm volatile("" ::: "memory");
  // the 64 folded partials of this thread's sum, all requested at once, added in bn_sum_partials' order (16 lanes of four, then
  // the 16 lane sums): bit-identical to k_bn_fold + k_bn_finalize / k_bn_bwd_finalize
  double s = 0;
  if (live) {
    float v[BN_FOLD];
#pragma unroll
    for (int j = 0; j < BN_FOLD; ++j) v[j] = __hip_atomic_load(folded + (int64_t)j * 2 * Cp + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int bl = 0; bl < 16; ++bl) {
      double lane_sum = 0;
#pragma unroll
      for (int j = bl; j < BN_FOLD; j += 16) lane_sum += (double)v[j];
      s += lane_sum;
    }
  }
  sums[half][cl] = s;
  __syncthreads();
  if (half == 0 && live) {
    if (BWD)
      bn_bwd_finalize_col(c, sums[0][cl], sums[1][cl], a.N, a.C, a.save_mean, a.save_invstd, a.weight, a.bias, a.d_weight, a.d_bias, a.out,
                          a.accumulate);
    else
      bn_finalize_col(a.col0 + c, sums[0][cl], sums[1][cl], a.N, a.C, a.save_mean, a.save_invstd, a.running_mean, a.running_var, a.weight,
                      a.bias, a.eps, a.momentum, 1, a.out);
  }
}
}  // extern "C++"
constexpr int BN_TICKET_WORD = 32;  // first ticket word in the control block (SCN_SK_STATE_BYTES): the words the split launches'
                                    // tickets use -- launches of one stream never overlap and each leaves them zero

static int bn_forward_impl(const float *x, float *y, float *save_mean, float *save_invstd,
                           float *running_mean, float *running_var, const float *weight,
                           const float *bias, double eps, double momentum, int train, double leakiness,
                           int64_t N, int C, void *temp, int nseg, const int32_t *seg_ncols,
                           const float *const *seg_partial, const int32_t *seg_nblk, hipStream_t stream, void *state = nullptr) {
  SCN_REQUIRE(C > 0 && N >= 0, "scn_bn_forward: bad shape");
  float *partial = (float *)temp;
  float *wb = partial + (size_t)BN_MAX_BLOCKS * 2 * C;
  const bool v4 = (C % 4) == 0;
  if (train && nseg > 0) {
    // statistics handed over by the producing kernels: only the finalize runs, once per segment of planes
    SCN_REQUIRE(N > 0, "scn_bn_forward: training batch norm over zero active sites");
    int col0 = 0;
    for (int i = 0; i < nseg; ++i) {
      SCN_REQUIRE(seg_ncols[i] > 0 && seg_nblk[i] > 0 && seg_partial[i] != nullptr, "scn_bn_forward_ext: bad segment");
      const float *pp = seg_partial[i];
      int pn = seg_nblk[i];
      if (pn > 4 * BN_FOLD && state != nullptr) {  // many partials: fold them 64-fold and finalize, one launch (k_bn_fold_finalize)
        float *folded = partial + (size_t)2 * BN_FOLD * col0;
        BnFinArgs a{N, seg_ncols[i], col0, C, 0, save_mean, save_invstd, running_mean, running_var, weight, bias, nullptr, nullptr, wb,
                    (float)eps, (float)momentum};
        k_bn_fold_finalize<false><<<dim3((seg_ncols[i] + 127) / 128, BN_FOLD), 256, 0, stream>>>(pp, pn, folded,
                                                                                                  (unsigned *)state + BN_TICKET_WORD, a);
        SCN_CHECK_LAUNCH();
        col0 += seg_ncols[i];
        continue;
      }
      if (pn > 4 * BN_FOLD) {  // ... or, without a control block, in two (the workspace's partial area is free here)
        float *folded = partial + (size_t)2 * BN_FOLD * col0;
        k_bn_fold<<<dim3((2 * seg_ncols[i] + 255) / 256, BN_FOLD), 256, 0, stream>>>(pp, pn, seg_ncols[i], folded);
        SCN_CHECK_LAUNCH();
        pp = folded;
        pn = BN_FOLD;
      }
      k_bn_finalize<<<(seg_ncols[i] + 15) / 16, 256, 0, stream>>>(pp, pn, N, seg_ncols[i], col0, C,
                                                                  save_mean, save_invstd, running_mean, running_var,
                                                                  weight, bias, (float)eps, (float)momentum, 1, wb);
      SCN_CHECK_LAUNCH();
      col0 += seg_ncols[i];
    }
    SCN_REQUIRE(col0 == C, "scn_bn_forward_ext: the segments must cover all planes");
  } else if (train && v4 && N > 1 && N <= BN_SMALL_ROWS) {
    // few rows (coarse UNet levels): statistics, finalize and apply in ONE launch
    k_bn_small_fwd<<<(C + 15) / 16, 256, 0, stream>>>(x, y, N, C, save_mean, save_invstd, running_mean, running_var,
                                                      weight, bias, (float)eps, (float)momentum, (float)leakiness);
    SCN_CHECK_LAUNCH();
    return 0;
  } else {
    const int nblk = bn_blocks(N);
    if (train) {
      SCN_REQUIRE(N > 0, "scn_bn_forward: training batch norm over zero active sites");
      const int cg = v4 ? C / 4 : C;
      dim3 grid(nblk, (cg + 255) / 256);
      if (v4) k_bn_stats<4><<<grid, 256, 0, stream>>>(x, N, C, partial);
      else k_bn_stats<1><<<grid, 256, 0, stream>>>(x, N, C, partial);
      SCN_CHECK_LAUNCH();
    }
    k_bn_finalize<<<(C + 15) / 16, 256, 0, stream>>>(partial, nblk, N, C, 0, C, save_mean, save_invstd,
                                                      running_mean, running_var, weight, bias,
                                                      (float)eps, (float)momentum, train, wb);
    SCN_CHECK_LAUNCH();
  }
  if (N > 0) {
    const int64_t total = N * C;
    if (v4) k_bn_apply<4><<<grid_for(total / 4), 256, 0, stream>>>(x, y, total / 4, C, wb, (float)leakiness);
    else k_bn_apply<1><<<grid_for(total), 256, 0, stream>>>(x, y, total, C, wb, (float)leakiness);
    SCN_CHECK_LAUNCH();
  }
  return 0;
}

int scn_bn_forward(const float *x, float *y, float *save_mean, float *save_invstd,
                   float *running_mean, float *running_var, const float *weight,
                   const float *bias, double eps, double momentum, int train, double leakiness,
                   int64_t N, int C, void *temp, void *stream_) {
  return bn_forward_impl(x, y, save_mean, save_invstd, running_mean, running_var, weight, bias, eps, momentum, train,
                         leakiness, N, C, temp, 0, nullptr, nullptr, nullptr, as_stream(stream_));
}

int scn_bn_forward_ext(const float *x, float *y, float *save_mean, float *save_invstd,
                       float *running_mean, float *running_var, const float *weight,
                       const float *bias, double eps, double momentum, double leakiness,
                       int64_t N, int C, void *temp, int nseg, const int32_t *seg_ncols_host,
                       const float *const *seg_partial_host, const int32_t *seg_nblk_host, void *state, void *stream_) {
  SCN_REQUIRE(nseg >= 1 && nseg <= 8, "scn_bn_forward_ext: 1..8 segments");
  return bn_forward_impl(x, y, save_mean, save_invstd, running_mean, running_var, weight, bias, eps, momentum, 1,
                         leakiness, N, C, temp, nseg, seg_ncols_host, seg_partial_host, seg_nblk_host, as_stream(stream_), state);
}

int scn_bn_backward(const float *x, float *d_x, const float *y, float *d_out,
                    const float *save_mean, const float *save_invstd, const float *weight, const float *bias,
                    float *d_weight, float *d_bias, double leakiness, int64_t N, int C,
                    int flags, void *temp, void *stream_) {
  const int rewrite_d_out = flags & 1, accumulate = (flags >> 1) & 1, from_x = (flags >> 2) & 1;
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(C > 0 && N > 0, "scn_bn_backward: bad shape");
  SCN_REQUIRE(!(from_x && rewrite_d_out), "scn_bn_backward: SCN_BN_MASK_FROM_X leaves d_out untouched; it cannot be combined with SCN_BN_REWRITE_D_OUT");
  SCN_REQUIRE(from_x || y != nullptr, "scn_bn_backward: y is required unless SCN_BN_MASK_FROM_X is set");
  float *partial = (float *)temp;
  float *coef = partial + (size_t)BN_MAX_BLOCKS * 2 * C;
  const int nblk = bn_blocks(N);
  const bool v4 = (C % 4) == 0;
  const int cg = v4 ? C / 4 : C;
  dim3 grid(nblk, (cg + 255) / 256);
  const float leak = (float)leakiness;
  if (from_x && v4 && N <= BN_SMALL_ROWS) {  // few rows: one launch (see k_bn_small_bwd)
    k_bn_small_bwd<<<(C + 15) / 16, 256, 0, stream>>>(x, d_out, d_x, N, C, save_mean, save_invstd, weight, bias, d_weight,
                                                      d_bias, leak, accumulate);
    SCN_CHECK_LAUNCH();
    return 0;
  }
  float *d_masked = rewrite_d_out ? d_out : d_x;  // pass 2 reads element i of it before writing element i of d_x
  if (from_x) {
    if (v4) k_bn_bwd_stats<4, true><<<grid, 256, 0, stream>>>(x, y, d_out, nullptr, save_mean, save_invstd, weight, bias, N, C, leak, partial);
    else k_bn_bwd_stats<1, true><<<grid, 256, 0, stream>>>(x, y, d_out, nullptr, save_mean, save_invstd, weight, bias, N, C, leak, partial);
  } else {
    if (v4) k_bn_bwd_stats<4, false><<<grid, 256, 0, stream>>>(x, y, d_out, d_masked, save_mean, save_invstd, weight, bias, N, C, leak, partial);
    else k_bn_bwd_stats<1, false><<<grid, 256, 0, stream>>>(x, y, d_out, d_masked, save_mean, save_invstd, weight, bias, N, C, leak, partial);
  }
  SCN_CHECK_LAUNCH();
  k_bn_bwd_finalize<<<(C + 15) / 16, 256, 0, stream>>>(partial, nblk, N, C, save_mean, save_invstd, weight, bias,
                                                        d_weight, d_bias, coef, accumulate);
  SCN_CHECK_LAUNCH();
  const int64_t total = N * C;
  if (from_x) {
    if (v4) k_bn_bwd_apply<4, true><<<grid_for(total / 4), 256, 0, stream>>>(x, d_out, d_x, save_mean, total / 4, C, coef, leak);
    else k_bn_bwd_apply<1, true><<<grid_for(total), 256, 0, stream>>>(x, d_out, d_x, save_mean, total, C, coef, leak);
  } else {
    if (v4) k_bn_bwd_apply<4, false><<<grid_for(total / 4), 256, 0, stream>>>(x, d_masked, d_x, save_mean, total / 4, C, coef, leak);
    else k_bn_bwd_apply<1, false><<<grid_for(total), 256, 0, stream>>>(x, d_masked, d_x, save_mean, total, C, coef, leak);
  }
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_bn_backward_ext(const float *x, float *d_x, const float *d_out, const float *save_mean,
                        const float *save_invstd, const float *weight, const float *bias, float *d_weight,
                        float *d_bias, double leakiness, int64_t N, int C, int flags, void *temp,
                        const float *partial_in, int nblk_in, const float *d_x_add, int64_t ld_add, void *state, void *stream_) {
  const int accumulate = (flags >> 1) & 1;
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(C > 0 && N > 0, "scn_bn_backward_ext: bad shape");
  SCN_REQUIRE((flags & ~SCN_BN_ACCUMULATE_DPARAMS) == 0, "scn_bn_backward_ext: only SCN_BN_ACCUMULATE_DPARAMS may be set");
  SCN_REQUIRE(partial_in == nullptr || nblk_in > 0, "scn_bn_backward_ext: partial sums without blocks");
  SCN_REQUIRE(ld_add == 0 || (ld_add >= C && ld_add % 4 == 0 && ld_add < (1 << 24)), "scn_bn_backward_ext: bad ld_add");
  const int lda = (ld_add == C) ? 0 : (int)ld_add;
  float *partial = (float *)temp;
  float *coef = partial + (size_t)BN_MAX_BLOCKS * 2 * C;
  const bool v4 = (C % 4) == 0;
  const float leak = (float)leakiness;
  if (partial_in == nullptr && v4 && N <= BN_SMALL_ROWS) {
    k_bn_small_bwd<<<(C + 15) / 16, 256, 0, stream>>>(x, d_out, d_x, N, C, save_mean, save_invstd, weight, bias, d_weight,
                                                      d_bias, leak, accumulate, d_x_add, lda);
    SCN_CHECK_LAUNCH();
    return 0;
  }
  const float *pp = partial_in;
  int pn = nblk_in;
  if (pp == nullptr) {
    pn = bn_blocks(N);
    const int cg = v4 ? C / 4 : C;
    dim3 grid(pn, (cg + 255) / 256);
    if (v4) k_bn_bwd_stats<4, true><<<grid, 256, 0, stream>>>(x, nullptr, d_out, nullptr, save_mean, save_invstd, weight, bias, N, C, leak, partial);
    else k_bn_bwd_stats<1, true><<<grid, 256, 0, stream>>>(x, nullptr, d_out, nullptr, save_mean, save_invstd, weight, bias, N, C, leak, partial);
    SCN_CHECK_LAUNCH();
    pp = partial;
  } else if (pn > 4 * BN_FOLD && state != nullptr) {  // one partial per 128-row tile / 32-pair block of the producing launch: fold
    // them 64-fold and finalize, one launch (k_bn_fold_finalize)
    BnFinArgs a{N, C, 0, C, accumulate, const_cast<float *>(save_mean), const_cast<float *>(save_invstd), nullptr, nullptr, weight, bias,
                d_weight, d_bias, coef, 0.f, 0.f};
    k_bn_fold_finalize<true><<<dim3((C + 127) / 128, BN_FOLD), 256, 0, stream>>>(pp, pn, partial, (unsigned *)state + BN_TICKET_WORD, a);
    SCN_CHECK_LAUNCH();
    pn = 0;
  } else if (pn > 4 * BN_FOLD) {
    k_bn_fold<<<dim3((2 * C + 255) / 256, BN_FOLD), 256, 0, stream>>>(pp, pn, C, partial);
    SCN_CHECK_LAUNCH();
    pp = partial;
    pn = BN_FOLD;
  }
  if (pn > 0) {
    k_bn_bwd_finalize<<<(C + 15) / 16, 256, 0, stream>>>(pp, pn, N, C, save_mean, save_invstd, weight, bias, d_weight, d_bias,
                                                          coef, accumulate);
    SCN_CHECK_LAUNCH();
  }
  const int64_t total = N * C;
  if (v4) k_bn_bwd_apply<4, true><<<grid_for(total / 4), 256, 0, stream>>>(x, d_out, d_x, save_mean, total / 4, C, coef, leak, d_x_add, lda);
  else k_bn_bwd_apply<1, true><<<grid_for(total), 256, 0, stream>>>(x, d_out, d_x, save_mean, total, C, coef, leak, d_x_add, lda);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_input_forward(const float *feats, float *out, const int32_t *order,
                      const int32_t *site_start, int64_t N, int C, int average, void *stream_) {
  if (N <= 0) return 0;
  k_input_fwd<<<grid_for(N * C), 256, 0, as_stream(stream_)>>>(feats, out, order, site_start, N, C, average);
  SCN_CHECK_LAUNCH();
  return 0;
}
int scn_input_backward(float *d_feats, const float *d_out, const int32_t *site_of_row,
                       const int32_t *site_start, int64_t P, int C, int average, void *stream_) {
  if (P <= 0) return 0;
  k_input_bwd<<<grid_for(P * C), 256, 0, as_stream(stream_)>>>(d_feats, d_out, site_of_row, site_start, P, C, average);
  SCN_CHECK_LAUNCH();
  return 0;
}
int scn_output_forward(const float *in, float *out, const int32_t *site_of_row, int64_t P, int C,
                       void *stream_) {
  if (P <= 0) return 0;
  if (C % 4 == 0) k_output_fwd<4><<<grid_for(P * (C / 4)), 256, 0, as_stream(stream_)>>>(in, out, site_of_row, P, C);
  else k_output_fwd<1><<<grid_for(P * C), 256, 0, as_stream(stream_)>>>(in, out, site_of_row, P, C);
  SCN_CHECK_LAUNCH();
  return 0;
}
int scn_output_backward(float *d_in, const float *d_out, const int32_t *order,
                        const int32_t *site_start, int64_t N, int C, void *stream_) {
  if (N <= 0) return 0;
  if (C % 4 == 0) k_output_bwd<4><<<grid_for(N * (C / 4)), 256, 0, as_stream(stream_)>>>(d_in, d_out, order, site_start, N, C);
  else k_output_bwd<1><<<grid_for(N * C), 256, 0, as_stream(stream_)>>>(d_in, d_out, order, site_start, N, C);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_column_sum(const float *x, float *out, int64_t rows, int C, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)C * 4, stream));
  if (rows <= 0) return 0;
  int blocks = (int)((rows + 255) / 256);
  if (blocks > 512) blocks = 512;
  k_column_sum<<<blocks, 256, 0, stream>>>(x, out, rows, C);
  SCN_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
