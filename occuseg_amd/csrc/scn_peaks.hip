// Measured peaks of the box the bench runs on (libscn_hip_tuning.so only; tools/peaks.py, bench.py's roofline):
//   * f32 MFMA: every SIMD issues independent v_mfma_f32_16x16x4_f32 chains from registers, nothing else;
//   * HBM stream: float4 grid-stride copy of a buffer far larger than the 256 MiB Infinity Cache.
// "verify both on the box with a stream/MFMA microbench, report measured peaks".
#include "scn_common.h"

namespace scn {
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_peak_mfma(float *sink, int iters) {
  floatx4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  const float a = (float)threadIdx.x * 1e-9f, b = (float)blockIdx.x * 1e-9f;
  for (int it = 0; it < iters; it += 8) {  // 64 MFMAs per trip: the loop branch is noise
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  floatx4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) s += acc[i];
  if (s.x == 123.456f) sink[0] = s.x + s.y + s.z + s.w;  // never true: keeps the chains alive
}

// one block = 256 threads x 4 float4 each, four independent loads in flight per thread, no loop
__global__ void __launch_bounds__(256) k_peak_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  if (i + 768 < n4) {
    const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
  }
}
// read-only stream: the sum keeps the loads alive, one store per block
__global__ void __launch_bounds__(256) k_peak_read(const float4 *__restrict__ src, float *__restrict__ sink, int64_t n4) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t base = (int64_t)blockIdx.x * 4096 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t i = base + j * 256;
    if (i < n4) {
      const float4 v = src[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[blockIdx.x & 63] = acc.x;
}
}  // namespace scn
using namespace scn;

extern "C" {

// TFLOP/s of back-to-back f32 MFMAs on all CUs (waves_per_simd waves per SIMD)
int scn_peak_mfma_f32(float *scratch_dev, int waves_per_simd, double *tflops_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  int dev = 0, cus = 0;
  SCN_CHECK_HIP(hipGetDevice(&dev));
  SCN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int iters = 20000;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  SCN_CHECK_HIP(hipEventCreate(&e0));
  SCN_CHECK_HIP(hipEventCreate(&e1));
  k_peak_mfma<<<blocks, 256, 0, stream>>>(scratch_dev, 100);  // warm-up (clocks)
  SCN_CHECK_HIP(hipEventRecord(e0, stream));
  k_peak_mfma<<<blocks, 256, 0, stream>>>(scratch_dev, iters);
  SCN_CHECK_HIP(hipEventRecord(e1, stream));
  SCN_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  SCN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 16 * 16 * 4;
  *tflops_host = flops / ((double)ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

// GB/s (bytes read + bytes written) of a float4 copy of `bytes` bytes, best of `reps`
int scn_peak_copy(const void *src, void *dst, int64_t bytes, int reps, double *gbs_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  const int64_t n4 = bytes / 16;
  hipEvent_t e0, e1;
  SCN_CHECK_HIP(hipEventCreate(&e0));
  SCN_CHECK_HIP(hipEventCreate(&e1));
  double best = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    SCN_CHECK_HIP(hipEventRecord(e0, stream));
    k_peak_copy<<<(unsigned)(n4 / 1024), 256, 0, stream>>>((const float4 *)src, (float4 *)dst, n4);
    SCN_CHECK_HIP(hipEventRecord(e1, stream));
    SCN_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SCN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = 2.0 * (double)n4 * 16.0 / ((double)ms * 1e-3) / 1e9;
    if (r > 0 && gbs > best) best = gbs;
  }
  *gbs_host = best;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

// GB/s of a read-only float4 stream over `bytes` bytes, best of `reps`
int scn_peak_read(const void *src, float *sink_dev, int64_t bytes, int reps, double *gbs_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  const int64_t n4 = bytes / 16;
  hipEvent_t e0, e1;
  SCN_CHECK_HIP(hipEventCreate(&e0));
  SCN_CHECK_HIP(hipEventCreate(&e1));
  double best = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    SCN_CHECK_HIP(hipEventRecord(e0, stream));
    k_peak_read<<<(unsigned)((n4 + 4095) / 4096), 256, 0, stream>>>((const float4 *)src, sink_dev, n4);
    SCN_CHECK_HIP(hipEventRecord(e1, stream));
    SCN_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SCN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = (double)n4 * 16.0 / ((double)ms * 1e-3) / 1e9;
    if (r > 0 && gbs > best) best = gbs;
  }
  *gbs_host = best;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

}  // extern "C"
