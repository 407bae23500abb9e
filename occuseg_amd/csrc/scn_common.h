// Shared helpers for libscn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/scn_hip.h"

namespace scn {

extern thread_local char g_err[512];

inline int fail(int code, const char *fmt, const char *a = "", long b = 0, long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return code;
}

#define SCN_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return scn::fail(100 + (int)e_, "HIP error %s at line %ld", hipGetErrorString(e_), __LINE__); \
  } while (0)

#define SCN_CHECK_LAUNCH() SCN_CHECK_HIP(hipGetLastError())

#define SCN_REQUIRE(cond, msg)                                         \
  do {                                                                 \
    if (!(cond)) return scn::fail(2, "%s (line %ld)", msg, __LINE__); \
  } while (0)

constexpr int WAVE = 64;

// stream-K launches of k_conv_tile4 (scn_conv.hip) and the tables scn_tile_passes builds for them (scn_rules.hip): the LAST
// 1 + fraction rounds of SCN_SK_SLICES (tile, column block) items are cut into SCN_SK_SLICES equal slices of passes -- the tile
// sequence into SCN_SK_SLICES / ncb parts, each walked once per column block --, the items in front of them are dealt whole
constexpr int SCN_SK_SLICES = 1024;
constexpr int SCN_SK_MAX_NCB = 6;  // column blocks (C_o / 64) a table exists for
// (tiles whose ncb items are dealt whole: a multiple of 512, so that tiles x ncb items divide into the XCDs' chunks of 64)
__host__ __device__ inline int64_t scn_sk_whole_tiles(int64_t tiles, int ncb, int static_schedule = 0) {
  const int64_t rounds = tiles * ncb / SCN_SK_SLICES;
  if (rounds < 2) return 0;
  // (static schedule: whole rounds of 1024 items = 1024 / ncb tiles each)
  return static_schedule ? (rounds - 1) * (SCN_SK_SLICES / ncb) : (rounds - 1) * SCN_SK_SLICES / ncb / 512 * 512;
}
constexpr int SCN_SK_CH = 64;  // slices per chunk of one XCD

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// 31-bit voxel key of the reference: ((z<<21)|(y<<10)|x) & 0x7FFFFFFF
// (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:80-81), widened with the sample index so
// that one global sort orders (sample, key) == the reference's per-sample ranks.
__device__ __forceinline__ uint64_t voxel_key(int x, int y, int z, int b) {
  uint32_t k = (((uint32_t)z << 21) | ((uint32_t)y << 10) | (uint32_t)x) & 0x7FFFFFFFu;
  return ((uint64_t)(uint32_t)b << 31) | (uint64_t)k;
}
__device__ __forceinline__ bool in_key_box(int x, int y, int z) {
  return (unsigned)x < 1024u && (unsigned)y < 2048u && (unsigned)z < 1024u;
}

__device__ __forceinline__ uint32_t hash_slot(uint64_t key, uint32_t mask) {
  key ^= key >> 33;
  key *= 0xff51afd7ed558ccdULL;
  key ^= key >> 33;
  key *= 0xc4ceb9fe1a85ec53ULL;
  key ^= key >> 33;
  return (uint32_t)key & mask;
}
constexpr uint64_t HASH_EMPTY = 0xFFFFFFFFFFFFFFFFULL;

// Order-preserving rank of the set flags inside a 256-thread block, using wave
// ballots + a 4-entry LDS prefix.  `wave_tot` must be a __shared__ int[4] per
// independent use between barriers.  Returns the rank of this thread's flag
// among the block's flags (valid when flag is set) and the block total.
__device__ __forceinline__ int block_rank256(bool flag, int *wave_tot, int &block_total) {
  const unsigned long long m = __ballot(flag);
  const int lane = lane_id();
  const int w = threadIdx.x >> 6;
  const int r = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[w] = __popcll(m);
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = wave_tot[i];
    base += (i < w) ? t : 0;
  }
  block_total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  __syncthreads();
  return base + r;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

}  // namespace scn
