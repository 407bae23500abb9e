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

// ---- MFMA issue probe (tools/mfma_probe.py): what a wave's instruction stream around its MFMAs costs.
// MODE: 0 two alternating accumulators, back to back   1 eight accumulators   2 as 0, A / B operands from 32 + 16 different
// registers   3 as 2 + one independent v_add between MFMAs   4 as 2 + three v_add   5 as 2 + one s_add   6 as 2 + a
// (never taken) scalar branch every 8 MFMAs   7 as 2 + one ds_read_b128 every 4 MFMAs (unused result)
// 8 as 2 + one buffer_load_dwordx4 with EXEC = 0 every 4 MFMAs   9 as 2 + one L2-hit buffer_load_dwordx4 every 4 MFMAs
template <int MODE>
__global__ void __launch_bounds__(256) k_mfma_probe(float *sink, const float *src, int iters, int never) {
  __shared__ float lds[4096];
  asm volatile("" ::: "a255");  // (mode 10 reads it)
  floatx4 acc[8];
  float a[32], b[16], f[4] = {0.f, 0.f, 0.f, 0.f}, g[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { g[i] = (float)(threadIdx.x * 3 + i) * 1e-9f; asm volatile("" : "+v"(g[i])); }
  floatx4 junk = {0.f, 0.f, 0.f, 0.f};
  int sc = never;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = (float)(threadIdx.x + i) * 1e-9f;
#pragma unroll
  for (int i = 0; i < 16; ++i) b[i] = (float)(blockIdx.x + i) * 1e-9f;
  lds[threadIdx.x] = a[0];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 1 << 20, 0x00020000);
  const unsigned voff = (threadIdx.x & 63) * 16;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {  // 64 MFMAs per trip
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      const int ai = MODE >= 2 ? (m & 31) : 0, bi = MODE >= 2 ? ((m >> 1) & 15) : 0;
      const int ci = MODE == 1 ? (m & 7) : (m & 1);
      // 28 / 29: the MFMA itself from inline asm with the accumulator in VGPRs / in AGPRs
      // 30-33: 28 / 29 + one L2-hit buffer_load_dwordx4 per 4 MFMAs (30 / 31) or per 2 MFMAs (32 / 33)
      if (MODE == 28 || MODE == 30 || MODE == 32) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[ci]) : "v"(a[ai]), "v"(b[bi]));
      else if (MODE == 29 || MODE == 31 || MODE == 33) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[ci]) : "v"(a[ai]), "v"(b[bi]));
      else acc[ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ai], b[bi], acc[ci], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 3) f[m & 3] += 1.0f;
      if (MODE == 4) { f[0] += 1.0f; f[1] += 2.0f; f[2] += 3.0f; }
      if (MODE == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc");
      if (MODE == 6 && (m & 7) == 7) { if (sc == 12345) sink[1] = 0.f; }
      if (MODE == 7 && (m & 3) == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(junk) : "v"(voff) : "memory");
      if (MODE == 8 && (m & 3) == 3)
        asm volatile("s_mov_b64 exec, 0\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_mov_b64 exec, -1" : "+v"(junk) : "v"(voff), "s"(rs) : "memory");
      if ((MODE == 32 || MODE == 33) && (m & 1) == 1)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(junk) : "v"(voff), "s"(rs) : "memory");
      if ((MODE == 9 || MODE == 30 || MODE == 31) && (m & 3) == 3)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(junk) : "v"(voff), "s"(rs) : "memory");
      // 10-17: one instruction per MFMA
      if (MODE == 10) asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(f[m & 3]));
      if (MODE == 11) asm volatile("v_mov_b64 %0, %1" : "=v"(*(double *)&junk) : "v"(*(double *)&a[(m & 15) * 2]));
      if (MODE == 12) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(f[m & 3]) : "v"(voff), "v"(a[m & 31]));
      if (MODE == 13) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(f[m & 3]) : "v"(voff), "v"(a[m & 31]), "v"(voff));
      if (MODE == 14) asm volatile("ds_add_f32 %0, %1" ::"v"(voff), "v"(a[m & 31]) : "memory");
      if (MODE == 15) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sc) : "v"(a[m & 31]));
      if (MODE == 16) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double *)&junk) : "v"(*(double *)&a[(m & 15) * 2]), "v"(*(double *)&b[(m & 7) * 2]));
      if (MODE == 17) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[m & 3]) : "v"(voff), "v"(a[m & 31]));
      if (MODE == 18) asm volatile("s_nop 0");
      // 20-23: the same VALU count as mode 13 (one per MFMA), clustered: 2 per 2, 4 per 4, 16 per 16, 64 per 64 MFMAs
      if (MODE >= 20 && MODE <= 23) {
        constexpr int CL = MODE == 20 ? 2 : MODE == 21 ? 4 : MODE == 22 ? 16 : 64;
        if ((m % CL) == CL - 1) {
#pragma unroll
          for (int u = 0; u < CL; ++u) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(f[u & 3]) : "v"(voff), "v"(a[(m + u) & 31]), "v"(voff));
        }
      }
      // 24: half as many (one per 2 MFMAs, single)   25: quarter (one per 4 MFMAs)   26: 16 per 64 clustered
      if (MODE == 24 && (m & 1)) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(f[m & 3]) : "v"(voff), "v"(a[m & 31]), "v"(voff));
      if (MODE == 25 && (m & 3) == 3) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(f[m & 3]) : "v"(voff), "v"(a[m & 31]), "v"(voff));
      if (MODE == 26 && m == 63) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(f[u & 3]) : "v"(voff), "v"(a[u & 31]), "v"(voff));
      }
      // 27: 16 v_pk_add_f32 per 64 clustered   28: ds_read_b128 x4 + 8 v_pk_add + ds_write_b128 x4 per 64 MFMAs (a tile update)
      if (MODE == 27 && m == 63) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double *)&f[(u & 1) * 2]) : "v"(*(double *)&a[(u & 15) * 2]), "v"(*(double *)&b[(u & 7) * 2]));
      }
      if (MODE == 19 && (m & 3) == 3) asm volatile("ds_write_b128 %0, %1" ::"v"(voff), "v"(acc[0]) : "memory");
      // 34-45 (round 4): what made row 3 (compiler-emitted f += 1.0f: 143 TF) so much cheaper than every inline-asm row (111 TF)?
      // the asm statement itself, the encoding (VOP2 / VOP3), the operand count, or that the asm rows all READ a register that
      // the MFMAs around them read too (a[], b[]).  g[] are registers no MFMA touches.
      if (MODE == 34) asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(f[m & 3]));                                 // row 3 as asm
      if (MODE == 35) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(f[m & 3]) : "v"(a[m & 31]));                 // + reads an MFMA operand
      if (MODE == 36) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(f[m & 3]) : "v"(g[m & 3]));                  // + reads a free register
      if (MODE == 37) asm volatile("v_add_f32_e64 %0, %1, %0" : "+v"(f[m & 3]) : "v"(g[m & 3]));                  // VOP3 encoding of 36
      if (MODE == 38) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(f[m & 3]) : "v"(g[m & 3]));                  // integer add
      if (MODE == 39) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(f[m & 3]) : "v"(g[m & 3]));              // VOP3 address form
      if (MODE == 40) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(*(double *)&f[(m & 1) * 2]) : "v"(*(double *)&g[(m & 1) * 2]));  // packed, free registers
      if (MODE == 41) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(f[m & 3]) : "v"(g[m & 3]), "v"(g[(m + 1) & 3]));            // three free sources
      if (MODE == 42) f[m & 3] += a[m & 31];                                                                       // compiler-emitted, reads an MFMA operand
      if (MODE == 43) { f[(m & 1) * 2] += g[(m & 1) * 2]; f[(m & 1) * 2 + 1] += g[(m & 1) * 2 + 1]; }              // compiler-emitted pair (v_pk_add_f32?)
      if (MODE == 44) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(f[m & 3]) : "v"(g[m & 3]));                       // plain move, free registers
      if (MODE == 45) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double *)&f[(m & 1) * 2]) : "v"(*(double *)&acc[2 + (m & 1)]), "v"(*(double *)&g[(m & 1) * 2]));  // packed, reads an idle accumulator tuple
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(junk) : "memory");
    if (MODE == 8 || MODE == 9 || MODE >= 30) asm volatile("s_waitcnt vmcnt(0)" ::"v"(junk) : "memory");
  }
  floatx4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) s += acc[i];
  s += junk;
  if (s.x + f[0] + f[1] + f[2] + f[3] + g[0] + g[1] + g[2] + g[3] + (float)sc == 123.456f) sink[0] = s.x + s.y + s.z + s.w;
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


// fraction of the f32-MFMA issue rate (one v_mfma_f32_16x16x4_f32 per 32 cycles and SIMD) that `waves_per_simd` waves per SIMD
// reach with stream MODE (see k_mfma_probe), at the clock the kernel itself measures
int scn_mfma_probe(float *scratch_dev, const float *src_dev, int mode, int waves_per_simd, double *tflops_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  int dev = 0, cus = 0;
  SCN_CHECK_HIP(hipGetDevice(&dev));
  SCN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int iters = 3000;
  const int blocks = cus * waves_per_simd;
  hipEvent_t e0, e1;
  SCN_CHECK_HIP(hipEventCreate(&e0));
  SCN_CHECK_HIP(hipEventCreate(&e1));
#define SCN_PROBE(M) case M: k_mfma_probe<M><<<blocks, 256, 0, stream>>>(scratch_dev, src_dev, n, 0); break
  for (int rep = 0; rep < 2; ++rep) {
    const int n = rep == 0 ? 50 : iters;
    if (rep == 1) SCN_CHECK_HIP(hipEventRecord(e0, stream));
    switch (mode) {
      SCN_PROBE(0); SCN_PROBE(1); SCN_PROBE(2); SCN_PROBE(3); SCN_PROBE(4); SCN_PROBE(5); SCN_PROBE(6); SCN_PROBE(7); SCN_PROBE(8); SCN_PROBE(9); SCN_PROBE(10); SCN_PROBE(11); SCN_PROBE(12); SCN_PROBE(13); SCN_PROBE(14); SCN_PROBE(15); SCN_PROBE(16); SCN_PROBE(17); SCN_PROBE(18); SCN_PROBE(19); SCN_PROBE(20); SCN_PROBE(21); SCN_PROBE(22); SCN_PROBE(23); SCN_PROBE(24); SCN_PROBE(25); SCN_PROBE(26); SCN_PROBE(27); SCN_PROBE(28); SCN_PROBE(29); SCN_PROBE(30); SCN_PROBE(31); SCN_PROBE(32); SCN_PROBE(33); SCN_PROBE(34); SCN_PROBE(35); SCN_PROBE(36); SCN_PROBE(37); SCN_PROBE(38); SCN_PROBE(39); SCN_PROBE(40); SCN_PROBE(41); SCN_PROBE(42); SCN_PROBE(43); SCN_PROBE(44); SCN_PROBE(45);
      default: return fail(2, "scn_mfma_probe: unknown mode%s", "");
    }
  }
#undef SCN_PROBE
  SCN_CHECK_HIP(hipEventRecord(e1, stream));
  SCN_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  SCN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4.0 * (double)iters * 64.0 * 2.0 * 16 * 16 * 4;
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
