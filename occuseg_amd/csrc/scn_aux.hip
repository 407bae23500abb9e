// The callers either side of the hot path, gfx950:
//   * per-site normals and their orientation class (normal-guided filters of the fork);
//   * ResolutionBasedScattering: the coarse site under each fine site, by key lookup;
//   * segment reductions over instance labels (the loss side: scatter_add / mean / std / max of torch_scatter).
// All HBM-bound integer / byte work: coalesced row reads, LDS partial tables, no MFMA.
#include "scn_common.h"

namespace scn {

// ------------------------------------------------------------ site normals ---
// normal[s] = normalize( (sum_j point_normal[p_j]) / cnt )  over the duplicates p_0 < p_1 < ... of site s, then
// OrientedFilter  (sparseconvnet/SCN/CUDA/IOLayers.cpp:36-65; Float3::normalize sparseconvnet/SCN/Metadata/Metadata.h:95-101;
// OrientedFilter sparseconvnet/SCN/Metadata/RectangularRegions.h:12-32: 0 = x dominant, 2 = y, 4 = z; ties in that order)
__global__ void k_site_normals(const float *__restrict__ pn, const int32_t *__restrict__ order,
                               const int32_t *__restrict__ site_start, int64_t N, float *__restrict__ sn,
                               int32_t *__restrict__ ori) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < N; s += (int64_t)gridDim.x * blockDim.x) {
    const int q0 = site_start[s], cnt = site_start[s + 1] - q0;
    float x = 0.f, y = 0.f, z = 0.f;
    for (int j = 0; j < cnt; ++j) {
      const float *p = pn + (int64_t)order[q0 + j] * 3;
      x += p[0]; y += p[1]; z += p[2];
    }
    if (cnt > 0) {
      x = __fdiv_rn(x, (float)cnt); y = __fdiv_rn(y, (float)cnt); z = __fdiv_rn(z, (float)cnt);
    }
    const float mag = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    if (!(mag < 1e-8f)) {
      const float inv = __fdiv_rn(1.f, mag);
      x = __fmul_rn(x, inv); y = __fmul_rn(y, inv); z = __fmul_rn(z, inv);
    }
    sn[s * 3 + 0] = x; sn[s * 3 + 1] = y; sn[s * 3 + 2] = z;
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    int face = 0;
    if (ax >= ay && ax >= az) face = 0;
    else if (ay >= ax && ay >= az) face = 2;
    else if (az >= ax && az >= ay) face = 4;
    ori[s] = face;
  }
}

// ------------------------------------------------------- coarse-site lookup ---
// out[q] = rank of key(query[q] / div) among the sorted unique site keys, or -1
// (ResolutionBasedScatteringCuda, sparseconvnet/SCN/Metadata/ConvolutionRules.h:327-342: insert the low-resolution points,
//  retrieve hr / stride; the retrieved value is the site id = key rank, 0xFFFFFFFF when absent)
__global__ void k_lookup_sites(const uint64_t *__restrict__ keys, int64_t N, const int32_t *__restrict__ query,
                               int ncols, int64_t Q, int div, int sample, int32_t *__restrict__ out) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += (int64_t)gridDim.x * blockDim.x) {
    const int32_t *c = query + q * ncols;
    const int x = c[0] / div, y = c[1] / div, z = c[2] / div;
    const int b = ncols > 3 ? c[3] : sample;
    int r = -1;
    if (in_key_box(x, y, z) && b >= 0) {
      const uint64_t key = voxel_key(x, y, z, b);
      int64_t lo = 0, hi = N;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
      }
      if (lo < N && keys[lo] == key) r = (int)lo;
    }
    out[q] = r;
  }
}

// ------------------------------------------------------- segment reductions ---
// out[seg[p], c] (+)= src[p, c] for p < P.  Each workgroup first reduces its rows into an LDS table [S_TILE][C] with
// LDS float atomics (labels repeat heavily: a scene has tens of instances), then adds the table to the result with
// one global atomic per touched (segment, column).  Segments outside the table's window go straight to global atomics.
constexpr int SEG_LDS_FLOATS = 8192;  // 32 KiB

template <int OP>  // 0 add, 1 max
__device__ __forceinline__ void seg_accumulate(float *dst, float v) {
  if (OP == 0) {
    unsafeAtomicAdd(dst, v);
  } else {
    // float max through the integer ordering trick (valid for non-NaN): positive floats order as ints, negative reversed
    if (v >= 0.f) atomicMax((int *)dst, __float_as_int(v));
    else atomicMin((unsigned int *)dst, __float_as_uint(v));
  }
}

template <int OP>
__global__ void __launch_bounds__(256)
    k_segment_reduce(const float *__restrict__ src, const int64_t *__restrict__ seg, int64_t P, int C, int S,
                     float *__restrict__ out, float init) {
  __shared__ float tab[SEG_LDS_FLOATS];
  const int s_tile = min(S, SEG_LDS_FLOATS / C);  // segments 0 .. s_tile-1 are staged in LDS
  for (int i = threadIdx.x; i < s_tile * C; i += 256) tab[i] = init;
  __syncthreads();
  const int64_t rows_per_blk = (P + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = min(P, r0 + rows_per_blk);
  for (int64_t i = r0 * C + threadIdx.x; i < r1 * C; i += 256) {
    const int64_t p = i / C;
    const int c = (int)(i - p * C);
    const int64_t s = seg[p];
    if (s < 0 || s >= S) continue;  // torch_scatter would fault; rows with a label outside [0, S) are skipped
    const float v = src[i];
    if (s < s_tile) seg_accumulate<OP>(tab + s * C + c, v);
    else seg_accumulate<OP>(out + s * C + c, v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < s_tile * C; i += 256) {
    const float v = tab[i];
    if (OP == 0 ? v != 0.f : v != init) seg_accumulate<OP>(out + i, v);
  }
}

// counts[s] += number of rows with label s
__global__ void __launch_bounds__(256)
    k_segment_count(const int64_t *__restrict__ seg, int64_t P, int S, float *__restrict__ counts) {
  __shared__ int tab[SEG_LDS_FLOATS];
  const int s_tile = min(S, SEG_LDS_FLOATS);
  for (int i = threadIdx.x; i < s_tile; i += 256) tab[i] = 0;
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    const int64_t s = seg[p];
    if (s < 0 || s >= S) continue;
    if (s < s_tile) atomicAdd(tab + s, 1);
    else unsafeAtomicAdd(counts + s, 1.f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < s_tile; i += 256)
    if (tab[i]) unsafeAtomicAdd(counts + i, (float)tab[i]);
}

// out[p, c] = table[seg[p], c] * (scale ? scale[seg[p]] : 1)   (the gather half: backward of add / mean, centring for std)
__global__ void k_segment_gather(const float *__restrict__ table, const int64_t *__restrict__ seg,
                                 const float *__restrict__ scale, int64_t P, int C, int S, float *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P * C; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / C;
    const int64_t s = seg[p];
    float v = 0.f;
    if (s >= 0 && s < S) v = table[s * C + (i - p * C)] * (scale ? scale[s] : 1.f);
    out[i] = v;
  }
}

// sum of squared deviations: out[seg[p], c] += (src[p,c] - mean[seg[p],c])^2
__global__ void __launch_bounds__(256)
    k_segment_sqdev(const float *__restrict__ src, const float *__restrict__ mean, const int64_t *__restrict__ seg,
                    int64_t P, int C, int S, float *__restrict__ out) {
  __shared__ float tab[SEG_LDS_FLOATS];
  const int s_tile = min(S, SEG_LDS_FLOATS / C);
  for (int i = threadIdx.x; i < s_tile * C; i += 256) tab[i] = 0.f;
  __syncthreads();
  const int64_t rows_per_blk = (P + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = min(P, r0 + rows_per_blk);
  for (int64_t i = r0 * C + threadIdx.x; i < r1 * C; i += 256) {
    const int64_t p = i / C;
    const int c = (int)(i - p * C);
    const int64_t s = seg[p];
    if (s < 0 || s >= S) continue;
    const float d = src[i] - mean[s * C + c];
    if (s < s_tile) unsafeAtomicAdd(tab + s * C + c, d * d);
    else unsafeAtomicAdd(out + s * C + c, d * d);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < s_tile * C; i += 256)
    if (tab[i] != 0.f) unsafeAtomicAdd(out + i, tab[i]);
}

static inline int aux_grid(int64_t n, int block = 256, int cap = 2048) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace scn
using namespace scn;

extern "C" {

int scn_site_normals(const float *point_normals, const int32_t *order, const int32_t *site_start, int64_t N,
                     float *site_normals, int32_t *site_ori, void *stream_) {
  if (N <= 0) return 0;
  k_site_normals<<<aux_grid(N), 256, 0, as_stream(stream_)>>>(point_normals, order, site_start, N, site_normals, site_ori);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_lookup_sites(const uint64_t *site_keys, int64_t N, const int32_t *query, int ncols, int64_t Q, int div,
                     int sample, int32_t *out, void *stream_) {
  SCN_REQUIRE(ncols == 3 || ncols == 4, "scn_lookup_sites: ncols must be 3 or 4");
  SCN_REQUIRE(div >= 1, "scn_lookup_sites: stride must be >= 1");
  if (Q <= 0) return 0;
  k_lookup_sites<<<aux_grid(Q), 256, 0, as_stream(stream_)>>>(site_keys, N, query, ncols, Q, div, sample, out);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_segment_reduce(const float *src, const int64_t *seg, int64_t P, int C, int S, int op, float *out,
                       void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(C >= 1 && C <= SEG_LDS_FLOATS && S >= 0 && (op == 0 || op == 1), "scn_segment_reduce: bad arguments");
  if (P <= 0 || S <= 0) return 0;
  const int blocks = aux_grid(P * C, 256 * 16, 1024);
  if (op == 0) k_segment_reduce<0><<<blocks, 256, 0, stream>>>(src, seg, P, C, S, out, 0.f);
  else k_segment_reduce<1><<<blocks, 256, 0, stream>>>(src, seg, P, C, S, out, -3.402823466e+38f);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_segment_count(const int64_t *seg, int64_t P, int S, float *counts, void *stream_) {
  if (P <= 0 || S <= 0) return 0;
  k_segment_count<<<aux_grid(P, 256 * 8, 1024), 256, 0, as_stream(stream_)>>>(seg, P, S, counts);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_segment_gather(const float *table, const int64_t *seg, const float *scale, int64_t P, int C, int S,
                       float *out, void *stream_) {
  if (P <= 0) return 0;
  k_segment_gather<<<aux_grid(P * C), 256, 0, as_stream(stream_)>>>(table, seg, scale, P, C, S, out);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_segment_sqdev(const float *src, const float *mean, const int64_t *seg, int64_t P, int C, int S, float *out,
                      void *stream_) {
  SCN_REQUIRE(C >= 1 && C <= SEG_LDS_FLOATS, "scn_segment_sqdev: bad plane count");
  if (P <= 0 || S <= 0) return 0;
  k_segment_sqdev<<<aux_grid(P * C, 256 * 16, 1024), 256, 0, as_stream(stream_)>>>(src, mean, seg, P, C, S, out);
  SCN_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
