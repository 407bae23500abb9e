// Voxel grids, voxel hash map and rulebooks on the device (gfx950).
//
// What the reference does with CUDPP radix-sort / cuckoo hash / 27+8 stream
// compactions and host round trips (sparseconvnet/SCN/CUDA/CUDPPWrapper.{hpp,cu},
// sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.{cpp,cu}) is done here with one
// global (sample,key) sort, an open-addressing hash map and order-preserving
// compaction by wavefront ballot + popcount prefix sums.  Rulebooks never leave
// the device.  Semantics: SURVEY.md Appendix A.
#include "scn_common.h"

#include <cstdlib>
#include <rocprim/rocprim.hpp>

namespace scn {
thread_local char g_err[512] = "";

// ------------------------------------------------------------------ keys ---
// coordinate element types: 0 = int32, 1 = int64, 2 = float32 (converted like torch's .long(): truncation toward 0)
template <int DT>
__device__ __forceinline__ void read_coords(const void *coords_, int64_t p, int ncols, int64_t &x, int64_t &y,
                                            int64_t &z, int64_t &b) {
  if (DT == 1) {
    const int64_t *c = (const int64_t *)coords_ + p * ncols;
    x = c[0]; y = c[1]; z = c[2]; b = ncols > 3 ? c[3] : 0;
  } else if (DT == 0) {
    const int32_t *c = (const int32_t *)coords_ + p * ncols;
    x = c[0]; y = c[1]; z = c[2]; b = ncols > 3 ? c[3] : 0;
  } else {
    const float *c = (const float *)coords_ + p * ncols;
    // NaN / huge values land outside the key box and raise the error flag below
    auto cv = [](float v) { return (v == v && fabsf(v) < 1e9f) ? (int64_t)v : (int64_t)-1; };
    x = cv(c[0]); y = cv(c[1]); z = cv(c[2]); b = ncols > 3 ? cv(c[3]) : 0;
  }
}

template <int DT>
__global__ void k_make_keys(const void *coords_, int ncols, int shift, int64_t P,
                            uint64_t *keys, uint32_t *idx, int *err) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P;
       p += (int64_t)gridDim.x * blockDim.x) {
    int64_t x, y, z, b;
    read_coords<DT>(coords_, p, ncols, x, y, z, b);
    x >>= shift; y >>= shift; z >>= shift;
    if (x < 0 || x >= 1024 || y < 0 || y >= 2048 || z < 0 || z >= 1024 || b < 0 || b >= 65536) {
      atomicOr(err, 1);
      x = y = z = b = 0;
    }
    keys[p] = voxel_key((int)x, (int)y, (int)z, (int)b);
    idx[p] = (uint32_t)p;
  }
}

// first-of-run flags over the sorted keys; per-block counts
__global__ void __launch_bounds__(256) k_mark_first(const uint64_t *keys, int64_t P, int *blk_cnt) {
  __shared__ int wt[4];
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool flag = q < P && (q == 0 || keys[q] != keys[q - 1]);
  int total;
  block_rank256(flag, wt, total);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = total;
}

// exclusive scan of each row of data[rows][n] in place; totals[row] = row sum
__device__ __forceinline__ void excl_scan_row(int *__restrict__ d, int n, int *__restrict__ total) {
  __shared__ int wsum[16];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  int carry = 0;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n ? d[i] : 0;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int t = wsum[j];
      wbase += j < w ? t : 0;
      tot += t;
    }
    if (i < n) d[i] = carry + wbase + x - v;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(1024) k_excl_scan_rows(int *data, int n, int *totals) {
  excl_scan_row(data + (int64_t)blockIdx.x * n, n, totals + blockIdx.x);
}

__global__ void __launch_bounds__(256)
    k_assign_sites(const uint64_t *keys, const uint32_t *order, int64_t P, const int *blk_base,
                   int32_t *site_of_row, int32_t *site_start) {
  __shared__ int wt[4];
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool flag = q < P && (q == 0 || keys[q] != keys[q - 1]);
  int total;
  const int r = block_rank256(flag, wt, total);
  if (q < P) {
    const int site = blk_base[blockIdx.x] + r + (flag ? 1 : 0) - 1;
    site_of_row[order[q]] = site;
    if (flag) site_start[site] = (int32_t)q;
    if (q == P - 1) site_start[site + 1] = (int32_t)P;
  }
}

template <int DT>
__global__ void k_site_coords(const void *coords_, int ncols, int shift, int64_t N,
                              const uint64_t *keys_sorted, const int32_t *order,
                              const int32_t *site_start, int4 *site_coords, uint64_t *site_keys) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < N;
       s += (int64_t)gridDim.x * blockDim.x) {
    const int q = site_start[s];
    const int64_t p = order[q];
    int64_t x, y, z, b;
    read_coords<DT>(coords_, p, ncols, x, y, z, b);
    site_coords[s] = make_int4((int)(x >> shift), (int)(y >> shift), (int)(z >> shift), (int)b);
    site_keys[s] = keys_sorted[q];
  }
}

// ---------------------------------------------------- grid chain counts ---
// Site counts of a whole chain of grids (the input grid and `levels - 1` 2/2 coarsenings of it) in ONE launch, so that a
// batch needs ONE host read-back for all of them instead of one per scale: every input row inserts its voxel key of
// scale 0, 1, ... into that scale's open-addressing set and stops at the first scale where the key was already there
// (whoever put it there goes on to the coarser scales: P + N_0 + N_1 + ... insertions in all, not levels x P).
template <int DT>
__global__ void __launch_bounds__(256)
    k_chain_count(const void *coords_, int ncols, int64_t P, int levels, uint64_t *tables, uint32_t mask, int *info) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool live = p < P;
  int64_t x = 0, y = 0, z = 0, b = 0;
  if (live) {
    read_coords<DT>(coords_, p, ncols, x, y, z, b);
    if (x < 0 || x >= 1024 || y < 0 || y >= 2048 || z < 0 || z >= 1024 || b < 0 || b >= 65536) {
      atomicOr(info + levels, 1);
      x = y = z = b = 0;
    }
  }
  for (int l = 0; l < levels; ++l) {
    if (live) {
      const uint64_t key = voxel_key((int)(x >> l), (int)(y >> l), (int)(z >> l), (int)b);
      uint64_t *t = tables + (size_t)l * ((size_t)mask + 1);
      uint32_t slot = hash_slot(key, mask);
      while (true) {
        const unsigned long long prev =
            atomicCAS((unsigned long long *)&t[slot], (unsigned long long)HASH_EMPTY, (unsigned long long)key);
        if (prev == HASH_EMPTY) break;
        if (prev == key) {
          live = false;
          break;
        }
        slot = (slot + 1) & mask;
      }
    }
    const uint64_t m = __ballot(live);
    if (m == 0) break;  // (wave-uniform)
    if (lane_id() == (int)__builtin_ctzll(m)) atomicAdd(info + l, __popcll(m));
  }
}

// ------------------------------------------------------------ hash map ---
__global__ void k_hash_insert(const uint64_t *site_keys, int64_t N, uint64_t *tkeys,
                              int32_t *tvals, uint32_t mask) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < N;
       s += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = site_keys[s];
    uint32_t slot = hash_slot(key, mask);
    while (true) {
      const unsigned long long prev =
          atomicCAS((unsigned long long *)&tkeys[slot], (unsigned long long)HASH_EMPTY,
                    (unsigned long long)key);
      if (prev == HASH_EMPTY) {
        tvals[slot] = (int32_t)s;
        break;
      }
      slot = (slot + 1) & mask;  // keys are unique: never equal to prev
    }
  }
}

__device__ __forceinline__ int hash_lookup(const uint64_t *tkeys, const int32_t *tvals,
                                           uint32_t mask, uint64_t key) {
  uint32_t slot = hash_slot(key, mask);
  while (true) {
    const uint64_t k = tkeys[slot];
    if (k == key) return tvals[slot];
    if (k == HASH_EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
}

// ----------------------------------------------- submanifold 3x3x3 rules ---
// pass 1: find the 27 neighbours of each site, write nbr[k][o], count per block.
// Sites are numbered in key order (sample, z, y, x) with x in the lowest bits, so the sites of one (sample, z, y) row with
// consecutive x have consecutive numbers: a row of three neighbours (x - 1, x, x + 1 at one (y + dy, z + dz)) needs ONE hash
// probe when its first member exists -- the other two are the next one or two sites, or absent, which a look at their
// coordinates tells -- a second probe when it does not, a third when the middle one is missing too (1.75 on the average at
// half-full neighbourhoods), and the site's own row none at all: ~14 probes per site instead of 26 (round 4: the probe kernel
// was 0.65 of the 2.14 ms of kernel time of an m64 batch's rule build, random 8-byte reads of a 16-24 MB table).
__global__ void __launch_bounds__(256)
    k_sub_probe(const int4 *coords, int64_t N, const uint64_t *tkeys, const int32_t *tvals,
                uint32_t mask, int32_t *nbr, int *blk_cnt, int nblk, const int32_t *site_ori, int only_class) {
  __shared__ int wcnt[4][27];
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = o < N;
  const int lane = lane_id(), w = threadIdx.x >> 6;
  int4 c = make_int4(0, 0, 0, 0);
  if (valid) c = coords[o];
  // only_class >= 0: keep only the output sites whose orientation class (0 / 2 / 4) is only_class
  const bool keep = valid && (only_class < 0 || site_ori[o] == only_class);
  // is site t the voxel (x, y, z) of this sample?
  auto is_at = [&](int64_t t, int x, int y, int z) __attribute__((always_inline)) {
    if (t < 0 || t >= N) return false;
    const int4 q = coords[t];
    return q.x == x && q.y == y && q.z == z && q.w == c.w;
  };
  auto lookup = [&](int x, int y, int z) __attribute__((always_inline)) {
    return in_key_box(x, y, z) ? hash_lookup(tkeys, tvals, mask, voxel_key(x, y, z, c.w)) : -1;
  };
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    // k = (dx+1)*9 + (dy+1)*3 + (dz+1): sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:61-74; r = (dy+1)*3 + (dz+1)
    const int dy = r / 3 - 1, dz = r % 3 - 1;
    int i[3] = {-1, -1, -1};
    if (valid) {
      const int y = c.y + dy, z = c.z + dz;
      if (r == 4) {  // the site's own row
        i[1] = (int)o;
        if (is_at(o - 1, c.x - 1, y, z)) i[0] = (int)o - 1;
        if (is_at(o + 1, c.x + 1, y, z)) i[2] = (int)o + 1;
      } else {
        const int t0 = lookup(c.x - 1, y, z);
        if (t0 >= 0) {
          i[0] = t0;
          if (is_at((int64_t)t0 + 1, c.x, y, z)) i[1] = t0 + 1;
          const int64_t nx = (int64_t)(i[1] >= 0 ? i[1] : t0) + 1;
          if (is_at(nx, c.x + 1, y, z)) i[2] = (int)nx;
        } else {
          const int t1 = lookup(c.x, y, z);
          if (t1 >= 0) {
            i[1] = t1;
            if (is_at((int64_t)t1 + 1, c.x + 1, y, z)) i[2] = t1 + 1;
          } else {
            i[2] = lookup(c.x + 1, y, z);
          }
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int k = d * 9 + r;
      int v = keep ? i[d] : -1;
      if (valid) nbr[(int64_t)k * N + o] = v;
      const unsigned long long m = __ballot(v >= 0);
      if (lane == 0) wcnt[w][k] = __popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int k = threadIdx.x;
    blk_cnt[(int64_t)k * nblk + blockIdx.x] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
  }
}

// Normal-guided filters: the pair found at offset i for output site o goes to list rot[ori(o)][i]
// (remap_rules_with_normal, sparseconvnet/SCN/Metadata/SubmanifoldConvolutionRules.h:213-245; the same table as
// `rot_index`, sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:8-15; ori = OrientedFilter in {0, 2, 4}).  Here: permute the
// probe results of each site (nbr_new[n][o] = nbr_old[inv(ori(o))[n]][o]) and count per NEW list, so that the ordinary
// order-preserving fill produces the remapped lists sorted by output site.
__constant__ int8_t c_rot_index[27 * 6] = {
    0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,
    24,25,26,21,22,23,18,19,20,15,16,17,12,13,14,9,10,11,6,7,8,3,4,5,0,1,2,
    6,7,8,15,16,17,24,25,26,3,4,5,12,13,14,21,22,23,0,1,2,9,10,11,18,19,20,
    18,19,20,9,10,11,0,1,2,21,22,23,12,13,14,3,4,5,24,25,26,15,16,17,6,7,8,
    2,11,20,5,14,23,8,17,26,1,10,19,4,13,22,7,16,25,0,9,18,3,12,21,6,15,24,
    18,9,0,21,12,3,24,15,6,19,10,1,22,13,4,25,16,7,20,11,2,23,14,5,26,17,8};

__global__ void __launch_bounds__(256)
    k_nbr_remap(const int32_t *nbr_old, int32_t *nbr_new, const int32_t *site_ori, int64_t N, int *blk_cnt, int nblk) {
  __shared__ int wcnt[4][27];
  __shared__ int8_t inv[6][27];
  if (threadIdx.x < 27 * 6) inv[threadIdx.x / 27][c_rot_index[threadIdx.x]] = (int8_t)(threadIdx.x % 27);
  __syncthreads();
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = o < N;
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int ori = valid ? site_ori[o] : 0;
#pragma unroll
  for (int n = 0; n < 27; ++n) {
    int i = -1;
    if (valid) {
      i = nbr_old[(int64_t)inv[ori][n] * N + o];
      nbr_new[(int64_t)n * N + o] = i;
    }
    const unsigned long long m = __ballot(i >= 0);
    if (lane == 0) wcnt[w][n] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int n = threadIdx.x;
    blk_cnt[(int64_t)n * nblk + blockIdx.x] = wcnt[0][n] + wcnt[1][n] + wcnt[2][n] + wcnt[3][n];
  }
}

// pass 2: order-preserving compaction of (in, out) pairs into rules[k]  (block `bi` of `nblk` over the N sites of one scale)
// `bases` (may be NULL): the tile-base table of the OUTPUT column (scn_tile_bases), which this pass knows anyway -- the first pair of
// list k with output site >= 32 g is the pair output site 32 g writes, or would write
template <int K>
__device__ __forceinline__ void rules_fill_body(const int32_t *__restrict__ nbr, int64_t N, const int *__restrict__ blk_base, int nblk,
                                                int2 *__restrict__ rules, int bi, int32_t *__restrict__ bases = nullptr) {
  __shared__ int wcnt[4][K];
  const int64_t o = (int64_t)bi * 256 + threadIdx.x;
  const bool valid = o < N;
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int in_k[K];
  int rk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    in_k[k] = valid ? nbr[(int64_t)k * N + o] : -1;
    const unsigned long long m = __ballot(in_k[k] >= 0);
    rk[k] = __popcll(m & lt);
    if (lane == 0) wcnt[w][k] = __popcll(m);
  }
  __syncthreads();
  const bool edge = bases != nullptr && valid && ((o & (SCN_TILE_GRANULE - 1)) == 0 || o == N - 1);
  const int n_gran = (int)((N + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (in_k[k] >= 0 || edge) {
      int base = blk_base[(int64_t)k * nblk + bi];
      for (int j = 0; j < w; ++j) base += wcnt[j][k];
      if (in_k[k] >= 0) rules[(int64_t)k * N + base + rk[k]] = make_int2(in_k[k], (int)o);
      if (edge) {
        if ((o & (SCN_TILE_GRANULE - 1)) == 0) bases[(int64_t)k * (n_gran + 1) + o / SCN_TILE_GRANULE] = base + rk[k];
        if (o == N - 1) bases[(int64_t)k * (n_gran + 1) + n_gran] = base + rk[k] + (in_k[k] >= 0 ? 1 : 0);
      }
    }
  }
}
template <int K>
__global__ void __launch_bounds__(256)
    k_rules_fill(const int32_t *nbr, int64_t N, const int *blk_base, int nblk, int2 *rules) {
  rules_fill_body<K>(nbr, N, blk_base, nblk, rules, (int)blockIdx.x);
}

// ---------------------------------------------------- strided 2/2 rules ---
// k = ((x&1)*2 + (y&1))*2 + (z&1): sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:549-554
__device__ __forceinline__ void str_count_body(const int4 *__restrict__ coords, int64_t N, int *__restrict__ blk_cnt, int nblk, int bi) {
  __shared__ int wcnt[4][8];
  const int64_t f = (int64_t)bi * 256 + threadIdx.x;
  const int lane = lane_id(), w = threadIdx.x >> 6;
  int koff = -1;
  if (f < N) {
    const int4 c = coords[f];
    koff = ((c.x & 1) * 2 + (c.y & 1)) * 2 + (c.z & 1);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const unsigned long long m = __ballot(koff == k);
    if (lane == 0) wcnt[w][k] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int k = threadIdx.x;
    blk_cnt[(int64_t)k * nblk + bi] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
  }
}
__global__ void __launch_bounds__(256)
    k_str_count(const int4 *coords, int64_t N, int *blk_cnt, int nblk) {
  str_count_body(coords, N, blk_cnt, nblk, (int)blockIdx.x);
}

// `coarse(f, c)` = the coarse site of fine site f at coordinates c
// `bases_fine` (may be NULL): the tile-base table of column 0 (the fine site, ascending in every list)
template <typename Coarse>
__device__ __forceinline__ void str_fill_body(const int4 *__restrict__ coords, int64_t N, Coarse coarse_of, const int *__restrict__ blk_base,
                                              int nblk, int2 *__restrict__ rules, int bi, int32_t *__restrict__ bases_fine = nullptr) {
  __shared__ int wcnt[4][8];
  const int64_t f = (int64_t)bi * 256 + threadIdx.x;
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int koff = -1, coarse = -1;
  if (f < N) {
    const int4 c = coords[f];
    koff = ((c.x & 1) * 2 + (c.y & 1)) * 2 + (c.z & 1);
    coarse = coarse_of(f, c);
  }
  int my_rank = 0;
  int rk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const unsigned long long m = __ballot(koff == k);
    rk[k] = __popcll(m & lt);
    if (koff == k) my_rank = rk[k];
    if (lane == 0) wcnt[w][k] = __popcll(m);
  }
  __syncthreads();
  if (koff >= 0) {
    int base = blk_base[(int64_t)koff * nblk + bi];
    for (int j = 0; j < w; ++j) base += wcnt[j][koff];
    rules[(int64_t)koff * N + base + my_rank] = make_int2((int)f, coarse);
  }
  if (bases_fine != nullptr && f < N && ((f & (SCN_TILE_GRANULE - 1)) == 0 || f == N - 1)) {
    const int n_gran = (int)((N + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int base = blk_base[(int64_t)k * nblk + bi];
      for (int j = 0; j < w; ++j) base += wcnt[j][k];
      if ((f & (SCN_TILE_GRANULE - 1)) == 0) bases_fine[(int64_t)k * (n_gran + 1) + f / SCN_TILE_GRANULE] = base + rk[k];
      if (f == N - 1) bases_fine[(int64_t)k * (n_gran + 1) + n_gran] = base + rk[k] + (koff == k ? 1 : 0);
    }
  }
}
__global__ void __launch_bounds__(256)
    k_str_fill(const int4 *coords, int64_t N, const int32_t *coarse_of_fine, const int *blk_base,
               int nblk, int2 *rules) {
  str_fill_body(coords, N, [&](int64_t f, const int4 &) { return coarse_of_fine[f]; }, blk_base, nblk, rules, (int)blockIdx.x);
}

// Pass prefix of the 128-row output tiles of one output column (scn_conv_forward's tile_passes): k_conv_tile4 walks the pairs
// of (tile, offset k) in passes of 32, so tile t costs L(t) = sum_k ceil(pairs_k(t) / 32) passes per 64-plane input chunk and
// output column block (an empty tile still costs its set-up: weight 1).  passes[t] = sum_{t' < t} max(1, L(t')): with it a
// persistent launch cuts the whole pass sequence into equal slices (stream-K) instead of dealing whole tiles.  One workgroup.
__global__ void __launch_bounds__(1024)
    k_tile_passes(const int32_t *__restrict__ bases, int K, int n_gran, int n_tiles, int32_t *__restrict__ passes) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int carry = 0;
  for (int base = 0; base < n_tiles; base += 1024) {
    const int t = base + tid;
    int wk = 0;
    if (t < n_tiles) {
      const int g0 = t * 4, g1 = min(g0 + 4, n_gran);
      for (int k = 0; k < K; ++k)
        wk += (bases[(int64_t)k * (n_gran + 1) + g1] - bases[(int64_t)k * (n_gran + 1) + g0] + 31) >> 5;
      wk = max(wk, 1);
    }
    int incl = wk;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    __syncthreads();  // (wsum of the previous round has been read)
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int v = wsum[i];
      before += i < wv ? v : 0;
      total += v;
    }
    if (t < n_tiles) passes[t] = carry + before + incl - wk;
    carry += total;
  }
  if (tid == 0) passes[n_tiles] = carry;
}

// Slice tables behind the prefix (one per column-block count ncb that divides SCN_SK_SLICES: 1, 2, 4): entry s = where part s
// of the sliced TILE sequence begins, as (tile, pass inside the tile in units of one pass per input chunk); SCN_SK_SLICES / ncb
// parts (each is walked once per column block, by ncb slices).  The sequence: tiles in order, tile t holding
// L(t) = passes[t + 1] - passes[t] units; the tiles [0, scn_sk_whole_tiles) are left out (their items are dealt whole).
// Boundaries within 4 units of a tile's ends move there (a piece of a few passes costs a tile set-up and a slab round trip).
__global__ void __launch_bounds__(256)
    k_sk_slices(const int32_t *__restrict__ passes, int n_tiles, int2 *__restrict__ tab) {
  const int ncb = blockIdx.y + 1;
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (ncb != 1 && ncb != 2 && ncb != 4) return;  // (slots 2, 4, 5 hold the static schedule's tables)
  const int parts = SCN_SK_SLICES / ncb;
  if (s > parts) return;
  int2 *out = tab + (size_t)(ncb - 1) * (SCN_SK_SLICES + 1);
  const int64_t na = scn_sk_whole_tiles(n_tiles, ncb);
  if (s == parts) { out[s] = make_int2(n_tiles, 0); return; }
  if (s == 0) { out[s] = make_int2((int)na, 0); return; }
  const int64_t X0 = passes[na], T = passes[n_tiles];
  const int64_t x = X0 + (T - X0) * s / parts;
  int lo = 0, hi = n_tiles;  // largest t in [0, n_tiles) with passes[t] <= x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (passes[mid] <= x) lo = mid; else hi = mid;
  }
  const int64_t l0 = passes[lo], L = passes[lo + 1] - l0;
  int64_t ps = x - l0;
  const int64_t d = L / 2 < 4 ? L / 2 : 4;
  if (ps < d) ps = 0;
  else if (L - ps <= d) { ps = 0; ++lo; }
  out[s] = make_int2(lo, (int)ps);
}

// The same tables for the STATIC schedule of a stream-K launch (table slots 2 / 4 / 5 for 1 / 2 / 4 column blocks): nothing is
// drawn, workgroup (XCD x, j) walks item (r * 8 + x) * 128 + j in round r of the whole items and then ONE slice, so a slice can be
// cut to fit its owner: part ts of the sliced tile sequence holds W - D(ts) units, D(ts) = what the owner's whole items hold and
// W = the mean of D + part over all owners.  (The owners of the ncb slices of one part walk the column blocks of the same tiles
// in every round: same D.)  One workgroup of 1024 threads per column-block count.
__global__ void __launch_bounds__(1024)
    k_sk_slices_static(const int32_t *__restrict__ passes, int n_tiles, int2 *__restrict__ tab) {
  const int ncb = blockIdx.y == 0 ? 1 : blockIdx.y == 1 ? 2 : 4, slot = blockIdx.y == 0 ? 2 : blockIdx.y == 1 ? 4 : 5;
  const int parts = SCN_SK_SLICES / ncb, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int2 *out = tab + (size_t)slot * (SCN_SK_SLICES + 1);
  __shared__ long long wsum[16];
  __shared__ long long tot_s;
  const int64_t na = scn_sk_whole_tiles(n_tiles, ncb, 1);
  const int rounds = (int)(na * ncb / SCN_SK_SLICES);
  const int64_t X0 = passes[na], R = passes[n_tiles] - X0;
  // D of the owner of part tid (slice s = tid * ncb)
  long long D = 0;
  if (tid < parts) {
    const int s = tid * ncb, q = s / SCN_SK_CH, i = s % SCN_SK_CH, x = q % 8, j = (q / 8) * SCN_SK_CH + i;
    for (int r = 0; r < rounds; ++r) {
      const int tile = ((r * 8 + x) * 128 + j) / ncb;
      D += passes[tile + 1] - passes[tile];
    }
  }
  auto block_sum_scan = [&](long long v, long long &excl) {  // exclusive scan over the 1024 threads; returns the total
    long long incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const long long y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    long long before = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      const long long t = wsum[w];
      before += w < wv ? t : 0;
      total += t;
    }
    excl = before + incl - v;
    return total;
  };
  long long dummy;
  const long long Dtot = block_sum_scan(D, dummy);
  // W = (sum D + R) / parts; share of part tid = W - D, at least one unit's worth so that boundaries stay ordered
  const long long Wn = Dtot + R;  // (numerator; shares are kept times `parts` to stay in integers)
  long long share = tid < parts ? Wn - D * parts : 0;
  if (tid < parts && share < 1) share = 1;
  long long excl;
  const long long stot = block_sum_scan(share, excl);
  if (tid == 0) tot_s = stot;
  __syncthreads();
  if (tid >= parts) return;
  if (tid == 0) {  // the two ends (parts may be 1024 = the block size: no thread of its own for the last entry)
    out[0] = make_int2((int)na, 0);
    out[parts] = make_int2(n_tiles, 0);
    return;
  }
  // (128-bit products avoided: excl <= stot ~ parts * (sum D + R) < 2^40, R < 2^31)
  const int64_t x = X0 + (int64_t)((double)excl / (double)tot_s * (double)R + 0.5);
  int lo = 0, hi = n_tiles;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (passes[mid] <= x) lo = mid; else hi = mid;
  }
  if (lo < (int)na) lo = (int)na;
  const int64_t l0 = passes[lo], L = passes[lo + 1] - l0;
  int64_t ps = x - l0;
  if (ps < 0) ps = 0;
  const int64_t d = L / 2 < 4 ? L / 2 : 4;
  if (ps < d) ps = 0;
  else if (L - ps <= d) { ps = 0; ++lo; }
  out[tid] = make_int2(lo, (int)ps);
}

// ------------------------------------------------------------ tile bases ---
__device__ __forceinline__ void tile_bases_body(const int32_t *__restrict__ rules, const int32_t *__restrict__ counts, int K, int64_t stride,
                                                int out_col, int n_gran, int32_t *__restrict__ bases, int64_t t) {
  const int64_t total = (int64_t)K * (n_gran + 1);
  if (t >= total) return;
  const int k = (int)(t / (n_gran + 1));
  const int g = (int)(t % (n_gran + 1));
  const int target = g * SCN_TILE_GRANULE;
  const int32_t *col = rules + ((int64_t)k * stride) * 2 + out_col;
  int lo = 0, hi = counts[k];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[(int64_t)mid * 2] < target) lo = mid + 1; else hi = mid;
  }
  bases[t] = lo;
}
__global__ void k_tile_bases(const int32_t *rules, const int32_t *counts, int K, int64_t stride,
                             int out_col, int n_gran, int32_t *bases) {
  tile_bases_body(rules, counts, K, stride, out_col, n_gran, bases, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// ------------------------------------------------------ input rule table ---
__global__ void k_max_active(const int32_t *site_start, int64_t N, int32_t *out) {
  int m = 0;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < N;
       s += (int64_t)gridDim.x * blockDim.x)
    m = max(m, site_start[s + 1] - site_start[s]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off));
  if (lane_id() == 0) atomicMax(out, m);
}

__global__ void k_input_table(const int32_t *order, const int32_t *site_start, int64_t N,
                              int max_active, int32_t *table) {
  const int wdt = 1 + max_active;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * wdt;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / wdt;
    const int j = (int)(t % wdt);
    const int q0 = site_start[s], cnt = site_start[s + 1] - q0;
    int v;
    if (j == 0) v = cnt;
    else v = (j - 1 < cnt) ? order[q0 + j - 1] : 0;
    table[t] = v;
  }
}


// ------------------------------------------------ occupancy bitmaps (round 6) ---
// include/scn_hip.h, "the same batch from OCCUPANCY BITMAPS".  The table the kernels read lies at byte 0 of the buffer (written by
// k_put from its by-value argument: no host-to-device copy whose source could go out of scope), the counters behind it.
struct OccDev {
  int32_t levels, nb, ox, oy, oz, ob, pad0, pad1;
  int32_t nx[SCN_BATCH_MAX_LEVELS], ny[SCN_BATCH_MAX_LEVELS], nz[SCN_BATCH_MAX_LEVELS], wx[SCN_BATCH_MAX_LEVELS];
  int64_t woff[SCN_BATCH_MAX_LEVELS + 1];
};
// What the multi-scale launches of scn_batch_build_occ read (byte OCC_HDR_BATCH of the buffer; offsets are bytes into the arena)
struct BatchDev {
  int32_t L, pad;
  int32_t N[SCN_BATCH_MAX_LEVELS];
  int32_t blk0[SCN_BATCH_MAX_LEVELS + 1];                               // first 256-site block of every scale in those launches
  int32_t sub_counts[SCN_BATCH_MAX_LEVELS], str_counts[SCN_BATCH_MAX_LEVELS];  // list lengths: index into `small` (-1: not built)
  int64_t coords[SCN_BATCH_MAX_LEVELS], keys[SCN_BATCH_MAX_LEVELS], sub_rules[SCN_BATCH_MAX_LEVELS], str_rules[SCN_BATCH_MAX_LEVELS];
  int64_t nbr[SCN_BATCH_MAX_LEVELS], sub_blk[SCN_BATCH_MAX_LEVELS], str_blk[SCN_BATCH_MAX_LEVELS];  // (scratch)
  int64_t sub_bases[SCN_BATCH_MAX_LEVELS], str_bases_fine[SCN_BATCH_MAX_LEVELS];  // tile-base tables the fill launches write themselves
  int64_t small;
};
struct TileDesc {
  int64_t rules, bases, stride, first;  // (first: the launch's first thread of this table)
  int32_t counts, K, out_col, n_gran;
};
struct TileDescs {
  int32_t n, pad;
  int64_t threads;
  TileDesc d[3 * SCN_BATCH_MAX_LEVELS];
};
constexpr int OCC_HDR_BATCH = 512, OCC_HDR_TILES = 2048, OCC_HDR_BYTES = 4096;
constexpr int OCC_INFO_START = 0;     // info[0 .. levels]: occupied voxels in front of every scale's first word ([levels] = all)
constexpr int OCC_INFO_OVERFLOW = 32;  // info[32]: != 0 = a voxel with more than SCN_OCC_SEGMENT_MAX input rows
static_assert(sizeof(OccDev) <= OCC_HDR_BATCH && OCC_HDR_BATCH + sizeof(BatchDev) <= OCC_HDR_TILES &&
                  OCC_HDR_TILES + sizeof(TileDescs) <= OCC_HDR_BYTES, "occupancy header");

template <typename T>
__global__ void k_put(T v, T *dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}
__device__ __forceinline__ int ml_level(const BatchDev *__restrict__ b, int blk) {
  int l = 0;
  while (l + 1 < b->L && blk >= b->blk0[l + 1]) ++l;
  return l;
}

// mm[0..3] = max of (-x, -y, -z, -b), mm[4..7] = max of (x, y, z, b), mm[8] |= 1 for a coordinate outside the key box; all nine
// start as 0x80808080 (one memset)
template <int DT>
__global__ void __launch_bounds__(256) k_occ_bounds(const void *coords_, int ncols, int64_t P, int *mm) {
  int lo[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN}, hi[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
  bool bad = false;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    int64_t x, y, z, b;
    read_coords<DT>(coords_, p, ncols, x, y, z, b);
    if (x < 0 || x >= 1024 || y < 0 || y >= 2048 || z < 0 || z >= 1024 || b < 0 || b >= 65536) {
      bad = true;
      continue;
    }
    const int v[4] = {(int)x, (int)y, (int)z, (int)b};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      lo[a] = max(lo[a], -v[a]);
      hi[a] = max(hi[a], v[a]);
    }
  }
  // one atomic per value and BLOCK (a few hundred blocks: per wave of a grid that covers the rows, 65,000 atomics on eight
  // addresses took 0.7 ms)
  __shared__ int red[4][9];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = max(lo[a], __shfl_down(lo[a], off));
      hi[a] = max(hi[a], __shfl_down(hi[a], off));
    }
    if (lane_id() == 0) {
      red[threadIdx.x >> 6][a] = lo[a];
      red[threadIdx.x >> 6][4 + a] = hi[a];
    }
  }
  const bool any_bad = __ballot(bad) != 0;
  if (lane_id() == 0) red[threadIdx.x >> 6][8] = any_bad ? 1 : 0;
  __syncthreads();
  if (threadIdx.x < 8) {
    const int v = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
    if (v != INT_MIN) atomicMax(mm + threadIdx.x, v);
  } else if (threadIdx.x == 8 && (red[0][8] | red[1][8] | red[2][8] | red[3][8])) {
    atomicOr(mm + 8, 1);
  }
}

__device__ __forceinline__ int64_t occ_word_index(const OccDev *__restrict__ d, int l, int rx, int ry, int rz, int rb) {
  return d->woff[l] + (((int64_t)rb * d->nz[l] + rz) * d->ny[l] + ry) * d->wx[l] + (rx >> 5);
}

// Bits of scale 0: one atomic OR per input row, nothing returned (fire and forget).  [Round 6, first form: every row walked the
// scales with RETURNING atomics and stopped where its bit was taken -- 130 us for 1.1 M rows, 260 with a look before the atomic.]
template <int DT>
__global__ void __launch_bounds__(256) k_occ_set(const void *coords_, int ncols, int64_t P, const uint64_t *__restrict__ buf, uint64_t *words) {
  const OccDev *__restrict__ d = (const OccDev *)buf;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  int64_t x, y, z, b;
  read_coords<DT>(coords_, p, ncols, x, y, z, b);
  const int rx = (int)x - d->ox, ry = (int)y - d->oy, rz = (int)z - d->oz, rb = (int)b - d->ob;
  // (scn_occ_bounds has seen every row: inside the box.  Anything else -- a coords tensor changed in between -- is dropped.)
  if (rx < 0 || rx >= d->nx[0] || ry < 0 || ry >= d->ny[0] || rz < 0 || rz >= d->nz[0] || rb < 0 || rb >= d->nb) return;
  uint32_t *w32 = (uint32_t *)words;
  atomicOr(w32 + 2 * occ_word_index(d, 0, rx, ry, rz, rb), 1u << (rx & 31));  // (little endian: the low half of the word)
}

// Bits of scale l + 1 from scale l: a coarse voxel is occupied when one of its 2 x 2 x 2 voxels is.  One thread per coarse word = 32
// coarse voxels of an x row = two words of each of four rows of scale l: OR them, OR neighbouring bits, keep the even ones.
__global__ void __launch_bounds__(256) k_occ_coarsen(const uint64_t *__restrict__ buf, uint64_t *__restrict__ words, int l) {
  const OccDev *__restrict__ d = (const OccDev *)buf;
  const int l1 = l + 1;
  const int wxc = d->wx[l1], nyc = d->ny[l1], nzc = d->nz[l1], wxf = d->wx[l], nyf = d->ny[l], nzf = d->nz[l];
  const int64_t Wc = d->woff[l1 + 1] - d->woff[l1];
  const int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (wi >= Wc) return;
  const int xw = (int)(wi % wxc);
  const int64_t row = wi / wxc;
  const int yc = (int)(row % nyc), zc = (int)((row / nyc) % nzc), rb = (int)(row / ((int64_t)nyc * nzc));
  const uint64_t *__restrict__ fw = words + d->woff[l];
  uint64_t f = 0;
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int yf = 2 * yc + dy, zf = 2 * zc + dz;
      if (yf < nyf && zf < nzf) {
        const uint64_t *r = fw + (((int64_t)rb * nzf + zf) * nyf + yf) * wxf;
        f |= (uint64_t)(uint32_t)r[2 * xw];
        if (2 * xw + 1 < wxf) f |= (uint64_t)(uint32_t)r[2 * xw + 1] << 32;
      }
    }
  uint64_t t = (f | (f >> 1)) & 0x5555555555555555ull;
  t = (t | (t >> 1)) & 0x3333333333333333ull;
  t = (t | (t >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  t = (t | (t >> 4)) & 0x00FF00FF00FF00FFull;
  t = (t | (t >> 8)) & 0x0000FFFF0000FFFFull;
  t = (t | (t >> 16)) & 0x00000000FFFFFFFFull;
  if (t) words[d->woff[l1] + wi] = t;  // (the buffer was cleared)
}

// population counts of 1024-word blocks (thread t: words 4 t .. 4 t + 3)
__global__ void __launch_bounds__(256) k_occ_count(const uint64_t *__restrict__ words, int *__restrict__ blk) {
  __shared__ int wt[4];
  const ulonglong2 *q = (const ulonglong2 *)(words + ((int64_t)blockIdx.x * 1024 + threadIdx.x * 4));
  const ulonglong2 a = q[0], b = q[1];
  int c = __popc((uint32_t)a.x) + __popc((uint32_t)a.y) + __popc((uint32_t)b.x) + __popc((uint32_t)b.y);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if (lane_id() == 0) wt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = wt[0] + wt[1] + wt[2] + wt[3];
}

// exclusive scan of one int per thread over a 256-thread block (wave scans + four LDS entries)
__device__ __forceinline__ int block_excl_scan256(int v, int *wt, int &total) {
  const int lane = lane_id(), w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off);
    if (lane >= off) x += y;
  }
  if (lane == 63) wt[w] = x;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) base += i < w ? wt[i] : 0;
  total = wt[0] + wt[1] + wt[2] + wt[3];
  __syncthreads();
  return base + x - v;
}

// high halves = occupied voxels in front of the word; info[l] = in front of scale l's first word
__global__ void __launch_bounds__(256) k_occ_base(const uint64_t *__restrict__ buf, uint64_t *__restrict__ words, const int *__restrict__ blk,
                                                  int *__restrict__ info) {
  __shared__ int wt[4];
  const OccDev *__restrict__ d = (const OccDev *)buf;
  const int64_t w0 = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  ulonglong2 *q = (ulonglong2 *)(words + w0);
  ulonglong2 a = q[0], b = q[1];
  const int c0 = __popc((uint32_t)a.x), c1 = __popc((uint32_t)a.y), c2 = __popc((uint32_t)b.x), c3 = __popc((uint32_t)b.y);
  int total;
  const int e = blk[blockIdx.x] + block_excl_scan256(c0 + c1 + c2 + c3, wt, total);
  const uint32_t p0 = (uint32_t)e, p1 = p0 + c0, p2 = p1 + c1, p3 = p2 + c2;
  a.x = (a.x & 0xFFFFFFFFull) | ((uint64_t)p0 << 32);
  a.y = (a.y & 0xFFFFFFFFull) | ((uint64_t)p1 << 32);
  b.x = (b.x & 0xFFFFFFFFull) | ((uint64_t)p2 << 32);
  b.y = (b.y & 0xFFFFFFFFull) | ((uint64_t)p3 << 32);
  q[0] = a;
  q[1] = b;
  const int levels = d->levels;
  for (int l = 1; l < levels; ++l) {  // (info[0] = 0 and info[levels] = the total come from the memset and the scan)
    const int64_t t = d->woff[l] - w0;
    if (t >= 0 && t < 4) info[OCC_INFO_START + l] = t == 0 ? p0 : t == 1 ? p1 : t == 2 ? p2 : p3;
  }
}

// coordinates and keys of every site of every scale: one thread per word
__global__ void __launch_bounds__(256) k_occ_sites(const uint64_t *__restrict__ buf, const uint64_t *__restrict__ words, int64_t W,
                                                   const int *__restrict__ info, char *__restrict__ arena) {
  const OccDev *__restrict__ d = (const OccDev *)buf;
  const BatchDev *__restrict__ out = (const BatchDev *)((const char *)buf + OCC_HDR_BATCH);
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= W) return;
  const uint64_t word = words[w];
  uint32_t bits = (uint32_t)word;
  if (bits == 0) return;
  int l = 0;
  const int levels = d->levels;
  while (l + 1 < levels && w >= d->woff[l + 1]) ++l;
  if (l >= out->L) return;  // (a scale the plan does not reach)
  const int64_t wi = w - d->woff[l];
  const int wx = d->wx[l], ny = d->ny[l], nz = d->nz[l];
  const int xw = (int)(wi % wx);
  const int64_t row = wi / wx;
  const int ry = (int)(row % ny), rz = (int)((row / ny) % nz), rb = (int)(row / ((int64_t)ny * nz));
  const int y = (d->oy >> l) + ry, z = (d->oz >> l) + rz, b = d->ob + rb, x0 = (d->ox >> l) + xw * 32;
  int r = (int)(word >> 32) - info[OCC_INFO_START + l];
  int4 *coords = (int4 *)(arena + out->coords[l]);
  uint64_t *keys = (uint64_t *)(arena + out->keys[l]);
  while (bits) {
    const int i = __builtin_ctz(bits);
    bits &= bits - 1;
    coords[r] = make_int4(x0 + i, y, z, b);
    keys[r] = voxel_key(x0 + i, y, z, b);
    ++r;
  }
}

// rank of voxel (rx, ry, rz, rb) -- relative to the box, at scale l -- whose bit is known to be set
__device__ __forceinline__ int occ_rank(const OccDev *__restrict__ d, const uint64_t *__restrict__ words, const int *__restrict__ info,
                                        int l, int rx, int ry, int rz, int rb) {
  const uint64_t word = words[occ_word_index(d, l, rx, ry, rz, rb)];
  return (int)(word >> 32) + __popc((uint32_t)word & ((1u << (rx & 31)) - 1u)) - info[OCC_INFO_START + l];
}

// input layer: the site of every input row, and its arrival number among the rows of its site
template <int DT>
__global__ void __launch_bounds__(256) k_occ_rows(const void *coords_, int ncols, int64_t P, const uint64_t *__restrict__ buf,
                                                  const uint64_t *__restrict__ words, const int *__restrict__ info,
                                                  int32_t *__restrict__ site_of_row, int32_t *__restrict__ slot, int *cnt) {
  const OccDev *__restrict__ d = (const OccDev *)buf;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  int64_t x, y, z, b;
  read_coords<DT>(coords_, p, ncols, x, y, z, b);
  const int rx = (int)x - d->ox, ry = (int)y - d->oy, rz = (int)z - d->oz, rb = (int)b - d->ob;
  if (rx < 0 || rx >= d->nx[0] || ry < 0 || ry >= d->ny[0] || rz < 0 || rz >= d->nz[0] || rb < 0 || rb >= d->nb) return;  // (see k_occ_set)
  const int s = occ_rank(d, words, info, 0, rx, ry, rz, rb);
  site_of_row[p] = s;
  slot[p] = atomicAdd(cnt + s, 1);
}

__global__ void __launch_bounds__(256) k_sum256(const int *__restrict__ v, int64_t n, int *__restrict__ blk) {
  __shared__ int wt[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int c = i < n ? v[i] : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if (lane_id() == 0) wt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = wt[0] + wt[1] + wt[2] + wt[3];
}

// site_start[i] = rows in front of site i; site_start[n] = P
__global__ void __launch_bounds__(256) k_starts256(const int *__restrict__ cnt, int64_t n, const int *__restrict__ blk,
                                                   int32_t *__restrict__ site_start, int P) {
  __shared__ int wt[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int total;
  const int e = blk[blockIdx.x] + block_excl_scan256(i < n ? cnt[i] : 0, wt, total);
  if (i < n) site_start[i] = e;
  if (i == n) site_start[n] = P;
}

__global__ void __launch_bounds__(256) k_occ_order(const int32_t *__restrict__ site_of_row, const int32_t *__restrict__ slot,
                                                   const int32_t *__restrict__ site_start, int64_t P, int32_t *__restrict__ order) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < P) order[site_start[site_of_row[p]] + slot[p]] = (int32_t)p;
}

// the rows of a site in ascending order (the atomics above hand out arrival numbers in any order): insertion sort of the short
// segments; a longer one raises the flag (include/scn_hip.h: the caller's business)
__global__ void __launch_bounds__(256) k_occ_order_sort(const int32_t *__restrict__ site_start, int64_t N, int32_t *__restrict__ order,
                                                        int *__restrict__ info) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= N) return;
  const int q0 = site_start[s], n = site_start[s + 1] - q0;
  if (n < 2) return;
  if (n > SCN_OCC_SEGMENT_MAX) {
    info[OCC_INFO_OVERFLOW] = 1;
    return;
  }
  int32_t *o = order + q0;
  for (int i = 1; i < n; ++i) {
    const int32_t key = o[i];
    int j = i - 1;
    while (j >= 0 && o[j] > key) {
      o[j + 1] = o[j];
      --j;
    }
    o[j + 1] = key;
  }
}

// small[2 l] = sites of scale l, small[2 l + 1] = flags (bit 1 at scale 0: a site's segment was not sorted)
__global__ void k_occ_small(const int *__restrict__ info, int levels, int32_t *__restrict__ small) {
  const int l = threadIdx.x;
  if (l >= levels) return;
  small[2 * l] = info[OCC_INFO_START + l + 1] - info[OCC_INFO_START + l];
  small[2 * l + 1] = (l == 0 && info[OCC_INFO_OVERFLOW]) ? 2 : 0;
}

// the 27 neighbours of every site of every scale from the bitmaps: one word per (y + dy, z + dz) row holds x - 1, x, x + 1 (two
// words when the site sits at a word's edge), and their ranks are consecutive.  Output as k_sub_probe.  One launch for all scales.
__global__ void __launch_bounds__(256)
    k_sub_probe_occ(const uint64_t *__restrict__ buf, const uint64_t *__restrict__ words, const int *__restrict__ info, char *__restrict__ arena) {
  __shared__ int wcnt[4][27];
  const OccDev *__restrict__ d = (const OccDev *)buf;
  const BatchDev *__restrict__ bd = (const BatchDev *)((const char *)buf + OCC_HDR_BATCH);
  const int l = ml_level(bd, (int)blockIdx.x);
  if (bd->sub_counts[l] < 0) return;
  const int bi = (int)blockIdx.x - bd->blk0[l], nblk = bd->blk0[l + 1] - bd->blk0[l];
  const int64_t N = bd->N[l];
  const int4 *__restrict__ coords = (const int4 *)(arena + bd->coords[l]);
  int32_t *__restrict__ nbr = (int32_t *)(arena + bd->nbr[l]);
  int *__restrict__ blk_cnt = (int *)(arena + bd->sub_blk[l]);
  const int64_t o = (int64_t)bi * 256 + threadIdx.x;
  const bool valid = o < N;
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int ny = d->ny[l], nz = d->nz[l], wx = d->wx[l], ls = info[OCC_INFO_START + l];
  const uint64_t *__restrict__ lw = words + d->woff[l];
  int rx = 0, ry = 0, rz = 0, rb = 0;
  if (valid) {
    const int4 c = coords[o];
    rx = c.x - (d->ox >> l); ry = c.y - (d->oy >> l); rz = c.z - (d->oz >> l); rb = c.w - d->ob;
  }
  const int p = rx & 31, wc = rx >> 5;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    // k = (dx+1)*9 + (dy+1)*3 + (dz+1): sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:61-74; r = (dy+1)*3 + (dz+1)
    const int dy = r / 3 - 1, dz = r % 3 - 1;
    int i[3] = {-1, -1, -1};
    const int y = ry + dy, z = rz + dz;
    if (valid && y >= 0 && y < ny && z >= 0 && z < nz) {
      const uint64_t *row = lw + (((int64_t)rb * nz + z) * ny + y) * wx;
      const uint64_t wd = row[wc];
      const uint32_t bits = (uint32_t)wd;
      const int base = (int)(wd >> 32) - ls;
      const int below = __popc(bits & ((1u << p) - 1u));  // occupied voxels of this word in front of x
      if ((bits >> p) & 1u) i[1] = base + below;
      if (p > 0) {
        if ((bits >> (p - 1)) & 1u) i[0] = base + below - 1;
      } else if (wc > 0) {
        const uint64_t wp = row[wc - 1];
        if ((uint32_t)wp >> 31) i[0] = (int)(wp >> 32) - ls + __popc((uint32_t)wp & 0x7FFFFFFFu);
      }
      if (p < 31) {
        if ((bits >> (p + 1)) & 1u) i[2] = base + below + (int)((bits >> p) & 1u);
      } else if (wc + 1 < wx) {
        const uint64_t wn = row[wc + 1];
        if ((uint32_t)wn & 1u) i[2] = (int)(wn >> 32) - ls;
      }
    }
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {
      const int k = dd * 9 + r;
      const int v = i[dd];
      if (valid) nbr[(int64_t)k * N + o] = v;
      const unsigned long long m = __ballot(v >= 0);
      if (lane == 0) wcnt[w][k] = __popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int k = threadIdx.x;
    blk_cnt[(int64_t)k * nblk + bi] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
  }
}

// the other per-scale launches of a batch, one launch each for all scales
__global__ void __launch_bounds__(256) k_str_count_ml(const uint64_t *__restrict__ buf, char *__restrict__ arena) {
  const BatchDev *__restrict__ bd = (const BatchDev *)((const char *)buf + OCC_HDR_BATCH);
  const int l = ml_level(bd, (int)blockIdx.x);
  if (bd->str_counts[l] < 0) return;
  str_count_body((const int4 *)(arena + bd->coords[l]), bd->N[l], (int *)(arena + bd->str_blk[l]), bd->blk0[l + 1] - bd->blk0[l],
                 (int)blockIdx.x - bd->blk0[l]);
}
// rows (blocks) scale x 35: 27 lists of the submanifold rulebook, 8 of the strided one
__global__ void __launch_bounds__(1024) k_scan_ml(const uint64_t *__restrict__ buf, char *__restrict__ arena) {
  const BatchDev *__restrict__ bd = (const BatchDev *)((const char *)buf + OCC_HDR_BATCH);
  const int l = (int)blockIdx.x / 35, j = (int)blockIdx.x % 35, nblk = bd->blk0[l + 1] - bd->blk0[l];
  int32_t *small = (int32_t *)(arena + bd->small);
  if (j < 27) {
    if (bd->sub_counts[l] >= 0) excl_scan_row((int *)(arena + bd->sub_blk[l]) + (int64_t)j * nblk, nblk, small + bd->sub_counts[l] + j);
  } else if (bd->str_counts[l] >= 0) {
    excl_scan_row((int *)(arena + bd->str_blk[l]) + (int64_t)(j - 27) * nblk, nblk, small + bd->str_counts[l] + j - 27);
  }
}
__global__ void __launch_bounds__(256) k_rules_fill_ml(const uint64_t *__restrict__ buf, char *__restrict__ arena) {
  const BatchDev *__restrict__ bd = (const BatchDev *)((const char *)buf + OCC_HDR_BATCH);
  const int l = ml_level(bd, (int)blockIdx.x);
  if (bd->sub_counts[l] < 0) return;
  rules_fill_body<27>((const int32_t *)(arena + bd->nbr[l]), bd->N[l], (const int *)(arena + bd->sub_blk[l]), bd->blk0[l + 1] - bd->blk0[l],
                      (int2 *)(arena + bd->sub_rules[l]), (int)blockIdx.x - bd->blk0[l], (int32_t *)(arena + bd->sub_bases[l]));
}
__global__ void __launch_bounds__(256) k_str_fill_ml(const uint64_t *__restrict__ buf, const uint64_t *__restrict__ words,
                                                     const int *__restrict__ info, char *__restrict__ arena) {
  const OccDev *__restrict__ d = (const OccDev *)buf;
  const BatchDev *__restrict__ bd = (const BatchDev *)((const char *)buf + OCC_HDR_BATCH);
  const int l = ml_level(bd, (int)blockIdx.x);
  if (bd->str_counts[l] < 0) return;
  const int l1 = l + 1;
  // the coarse site of a site: its voxel halved, looked up at the next scale
  auto coarse = [&](int64_t, const int4 &c) {
    return occ_rank(d, words, info, l1, (c.x >> 1) - (d->ox >> l1), (c.y >> 1) - (d->oy >> l1), (c.z >> 1) - (d->oz >> l1), c.w - d->ob);
  };
  str_fill_body((const int4 *)(arena + bd->coords[l]), bd->N[l], coarse, (const int *)(arena + bd->str_blk[l]), bd->blk0[l + 1] - bd->blk0[l],
                (int2 *)(arena + bd->str_rules[l]), (int)blockIdx.x - bd->blk0[l], (int32_t *)(arena + bd->str_bases_fine[l]));
}
__global__ void __launch_bounds__(256) k_tile_bases_ml(const uint64_t *__restrict__ buf, char *__restrict__ arena) {
  const BatchDev *__restrict__ bd = (const BatchDev *)((const char *)buf + OCC_HDR_BATCH);
  const TileDescs *__restrict__ td = (const TileDescs *)((const char *)buf + OCC_HDR_TILES);
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= td->threads) return;
  int i = 0;
  while (i + 1 < td->n && t >= td->d[i + 1].first) ++i;
  const TileDesc &q = td->d[i];
  tile_bases_body((const int32_t *)(arena + q.rules), (const int32_t *)(arena + bd->small) + q.counts, q.K, q.stride, q.out_col, q.n_gran,
                  (int32_t *)(arena + q.bases), t - q.first);
}

static inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

struct GridTemp {
  size_t keys_in, idx_in, blk_cnt, info, sort, total, sort_bytes;
};
// The 48-bit key sort.  rocPRIM's default Onesweep configuration for (u64, u32) on gfx950 compiles to a kernel with 80
// bytes of SCRATCH (private memory) per lane -- the only kernel of this library that had any.  8-bit digits, 256 x 12 keys
// per block: no scratch, six passes over the 48 bits; the rule build of an m64 batch alone takes 2.5 instead of 2.3 ms
// (tools/check_scratch.py keeps the library scratch-free).  History: for a day of round 3 this looked like the cure of GPU memory access faults seen under
// rocprofv3 (default sort 4 of 4 runs, this one 0 of 8 on the same box); the faults were a lifetime race of the side-stream
// dWeight kernels (SCN.py, _dw_side_end) and the sort's different workspace size merely moved which freed block the next
// rule build overwrote first.  SCN_SORT_ROCPRIM_DEFAULT=1 selects rocPRIM's default (A/B).
using SortNoScratch = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<256, 12>, 8>>;
static bool sort_rocprim_default() {
  static const bool v = [] {
    const char *e = getenv("SCN_SORT_ROCPRIM_DEFAULT");
    return e != nullptr && e[0] == '1';
  }();
  return v;
}
static hipError_t sort_keys(void *temp, size_t &temp_bytes, uint64_t *keys_in, uint64_t *keys_out, uint32_t *idx_in,
                            uint32_t *idx_out, size_t P, hipStream_t stream) {
  if (sort_rocprim_default())
    return rocprim::radix_sort_pairs<rocprim::default_config, uint64_t *, uint64_t *, uint32_t *, uint32_t *>(
        temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, P, 0, 47, stream, false);
  return rocprim::radix_sort_pairs<SortNoScratch, uint64_t *, uint64_t *, uint32_t *, uint32_t *>(
      temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, P, 0, 47, stream, false);
}
static int grid_temp_layout(int64_t P, GridTemp &t) {
  const int64_t nblk = (P + 255) / 256;
  size_t sort_bytes = 0;
  hipError_t e = sort_keys(nullptr, sort_bytes, nullptr, nullptr, nullptr, nullptr, (size_t)P, 0);
  if (e != hipSuccess) return 1;
  size_t off = 0;
  t.keys_in = off; off += align256((size_t)P * 8);
  t.idx_in = off; off += align256((size_t)P * 4);
  t.blk_cnt = off; off += align256((size_t)(nblk + 1) * 4);
  t.info = off; off += 256;
  t.sort = off; off += align256(sort_bytes);
  t.total = off;
  t.sort_bytes = sort_bytes;
  return 0;
}

}  // namespace scn

using namespace scn;

extern "C" {

int scn_abi_version(void) { return SCN_ABI_VERSION; }
const char *scn_last_error(void) { return scn::g_err; }

size_t scn_grid_build_temp_bytes(int64_t P) {
  GridTemp t;
  if (P <= 0 || grid_temp_layout(P, t)) return 0;
  return t.total;
}

// the launches of one grid build; `info` (device): [0] = N, [1] = error flag
static int grid_build_launches(const void *coords, int coords_dtype, int ncols, int shift, int64_t P, void *temp,
                               size_t temp_bytes, uint64_t *keys_sorted, int32_t *order, int32_t *site_of_row,
                               int32_t *site_start, int *info, hipStream_t stream) {
  SCN_REQUIRE(P > 0 && P < (int64_t)1 << 31, "scn_grid_build: P out of range");
  SCN_REQUIRE(ncols == 3 || ncols == 4, "scn_grid_build: ncols must be 3 or 4");
  SCN_REQUIRE(coords_dtype >= 0 && coords_dtype <= 2, "scn_grid_build: coords_dtype must be 0 (int32), 1 (int64) or 2 (float32)");
  GridTemp t;
  SCN_REQUIRE(grid_temp_layout(P, t) == 0, "scn_grid_build: rocprim size query failed");
  SCN_REQUIRE(temp_bytes >= t.total, "scn_grid_build: workspace too small");
  char *base = (char *)temp;
  uint64_t *keys_in = (uint64_t *)(base + t.keys_in);
  uint32_t *idx_in = (uint32_t *)(base + t.idx_in);
  int *blk_cnt = (int *)(base + t.blk_cnt);
  if (info == nullptr) info = (int *)(base + t.info);
  const int nblk = (int)((P + 255) / 256);
  SCN_CHECK_HIP(hipMemsetAsync(info, 0, 8, stream));
  if (coords_dtype == 1)
    k_make_keys<1><<<grid_for(P), 256, 0, stream>>>(coords, ncols, shift, P, keys_in, idx_in, info + 1);
  else if (coords_dtype == 0)
    k_make_keys<0><<<grid_for(P), 256, 0, stream>>>(coords, ncols, shift, P, keys_in, idx_in, info + 1);
  else
    k_make_keys<2><<<grid_for(P), 256, 0, stream>>>(coords, ncols, shift, P, keys_in, idx_in, info + 1);
  SCN_CHECK_LAUNCH();
  size_t sb = t.sort_bytes;
  SCN_CHECK_HIP(sort_keys(base + t.sort, sb, keys_in, keys_sorted, idx_in, (uint32_t *)order, (size_t)P, stream));
  k_mark_first<<<nblk, 256, 0, stream>>>(keys_sorted, P, blk_cnt);
  SCN_CHECK_LAUNCH();
  k_excl_scan_rows<<<1, 1024, 0, stream>>>(blk_cnt, nblk, info);
  SCN_CHECK_LAUNCH();
  k_assign_sites<<<nblk, 256, 0, stream>>>(keys_sorted, (const uint32_t *)order, P, blk_cnt,
                                          site_of_row, site_start);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_grid_build(const void *coords, int coords_dtype, int ncols, int shift, int64_t P,
                   void *temp, size_t temp_bytes, uint64_t *keys_sorted, int32_t *order,
                   int32_t *site_of_row, int32_t *site_start, int32_t *n_sites_host,
                   void *stream_) {
  hipStream_t stream = as_stream(stream_);
  if (int rc = grid_build_launches(coords, coords_dtype, ncols, shift, P, temp, temp_bytes, keys_sorted, order,
                                   site_of_row, site_start, nullptr, stream))
    return rc;
  GridTemp t;
  grid_temp_layout(P, t);
  int host_info[2] = {0, 0};
  SCN_CHECK_HIP(hipMemcpyAsync(host_info, (char *)temp + t.info, 8, hipMemcpyDeviceToHost, stream));
  SCN_CHECK_HIP(hipStreamSynchronize(stream));
  if (host_info[1])
    return fail(3, "scn_grid_build: coordinate outside the 31-bit key box (0<=x<1024, 0<=y<2048, 0<=z<1024) or bad sample index%s", "");
  *n_sites_host = host_info[0];
  return 0;
}

int scn_grid_build_async(const void *coords, int coords_dtype, int ncols, int shift, int64_t P,
                         void *temp, size_t temp_bytes, uint64_t *keys_sorted, int32_t *order,
                         int32_t *site_of_row, int32_t *site_start, int32_t *status_dev, void *stream_) {
  SCN_REQUIRE(status_dev != nullptr, "scn_grid_build_async: status_dev is required");
  return grid_build_launches(coords, coords_dtype, ncols, shift, P, temp, temp_bytes, keys_sorted, order, site_of_row,
                             site_start, (int *)status_dev, as_stream(stream_));
}

static int64_t chain_capacity(int64_t P) {
  int64_t cap = 1024;
  while (cap < 2 * P) cap <<= 1;
  return cap;
}

size_t scn_grid_chain_temp_bytes(int64_t P, int levels) {
  if (P <= 0 || levels < 1 || levels > SCN_GRID_CHAIN_MAX) return 0;
  return (size_t)levels * (size_t)chain_capacity(P) * 8 + 256;
}

int scn_grid_chain_counts(const void *coords, int coords_dtype, int ncols, int64_t P, int levels, void *temp,
                          size_t temp_bytes, int32_t *n_sites_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(P > 0 && P < (int64_t)1 << 30, "scn_grid_chain_counts: P out of range");
  SCN_REQUIRE(ncols == 3 || ncols == 4, "scn_grid_chain_counts: ncols must be 3 or 4");
  SCN_REQUIRE(coords_dtype >= 0 && coords_dtype <= 2, "scn_grid_chain_counts: coords_dtype must be 0 (int32), 1 (int64) or 2 (float32)");
  SCN_REQUIRE(levels >= 1 && levels <= SCN_GRID_CHAIN_MAX, "scn_grid_chain_counts: levels must be 1..SCN_GRID_CHAIN_MAX");
  SCN_REQUIRE(temp_bytes >= scn_grid_chain_temp_bytes(P, levels), "scn_grid_chain_counts: workspace too small");
  const int64_t cap = chain_capacity(P);
  uint64_t *tables = (uint64_t *)temp;
  int *info = (int *)((char *)temp + (size_t)levels * (size_t)cap * 8);  // [levels] counts, [levels] error flag
  SCN_CHECK_HIP(hipMemsetAsync(tables, 0xFF, (size_t)levels * (size_t)cap * 8, stream));
  SCN_CHECK_HIP(hipMemsetAsync(info, 0, 4 * (SCN_GRID_CHAIN_MAX + 1), stream));
  const int nblk = (int)((P + 255) / 256);
  if (coords_dtype == 1)
    k_chain_count<1><<<nblk, 256, 0, stream>>>(coords, ncols, P, levels, tables, (uint32_t)(cap - 1), info);
  else if (coords_dtype == 0)
    k_chain_count<0><<<nblk, 256, 0, stream>>>(coords, ncols, P, levels, tables, (uint32_t)(cap - 1), info);
  else
    k_chain_count<2><<<nblk, 256, 0, stream>>>(coords, ncols, P, levels, tables, (uint32_t)(cap - 1), info);
  SCN_CHECK_LAUNCH();
  if (n_sites_host == nullptr) return 0;  // (queued only: scn_grid_chain_fetch reads the counts back)
  return scn_grid_chain_fetch(temp, P, levels, n_sites_host, stream_);
}

int scn_grid_chain_fetch(const void *temp, int64_t P, int levels, int32_t *n_sites_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(P > 0 && levels >= 1 && levels <= SCN_GRID_CHAIN_MAX && n_sites_host != nullptr, "scn_grid_chain_fetch: bad arguments");
  const int *info = (const int *)((const char *)temp + (size_t)levels * (size_t)chain_capacity(P) * 8);
  int host_info[SCN_GRID_CHAIN_MAX + 1];
  SCN_CHECK_HIP(hipMemcpyAsync(host_info, info, 4 * (levels + 1), hipMemcpyDeviceToHost, stream));
  SCN_CHECK_HIP(hipStreamSynchronize(stream));
  if (host_info[levels])
    return fail(3, "scn_grid_chain_counts: coordinate outside the 31-bit key box (0<=x<1024, 0<=y<2048, 0<=z<1024) or bad sample index%s", "");
  for (int l = 0; l < levels; ++l) n_sites_host[l] = host_info[l];
  return 0;
}

int scn_grid_finalize(const void *coords, int coords_dtype, int ncols, int shift, int64_t P,
                      int64_t N, const uint64_t *keys_sorted, const int32_t *order,
                      const int32_t *site_start, int32_t *site_coords, uint64_t *site_keys,
                      void *stream_) {
  hipStream_t stream = as_stream(stream_);
  (void)P;
  if (N <= 0) return 0;
  if (coords_dtype == 1)
    k_site_coords<1><<<grid_for(N), 256, 0, stream>>>(coords, ncols, shift, N, keys_sorted, order,
                                                      site_start, (int4 *)site_coords, site_keys);
  else if (coords_dtype == 0)
    k_site_coords<0><<<grid_for(N), 256, 0, stream>>>(coords, ncols, shift, N, keys_sorted, order,
                                                      site_start, (int4 *)site_coords, site_keys);
  else
    k_site_coords<2><<<grid_for(N), 256, 0, stream>>>(coords, ncols, shift, N, keys_sorted, order,
                                                      site_start, (int4 *)site_coords, site_keys);
  SCN_CHECK_LAUNCH();
  return 0;
}

int64_t scn_hash_capacity(int64_t N) {
  int64_t cap = 1024;
  while (cap < 2 * N) cap <<= 1;
  return cap;
}

int scn_hash_build(const uint64_t *site_keys, int64_t N, uint64_t *table_keys,
                   int32_t *table_vals, int64_t capacity, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(capacity >= 2 * N && (capacity & (capacity - 1)) == 0 && capacity <= ((int64_t)1 << 31),
              "scn_hash_build: capacity must be a power of two >= 2N");
  SCN_CHECK_HIP(hipMemsetAsync(table_keys, 0xFF, (size_t)capacity * 8, stream));
  if (N > 0) {
    k_hash_insert<<<grid_for(N), 256, 0, stream>>>(site_keys, N, table_keys, table_vals,
                                                   (uint32_t)(capacity - 1));
    SCN_CHECK_LAUNCH();
  }
  return 0;
}

int scn_input_rules_max_active(const int32_t *site_start, int64_t N, int32_t *scratch_dev,
                               int32_t *max_active_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_CHECK_HIP(hipMemsetAsync(scratch_dev, 0, 4, stream));
  if (N > 0) {
    k_max_active<<<grid_for(N, 256, 1024), 256, 0, stream>>>(site_start, N, scratch_dev);
    SCN_CHECK_LAUNCH();
  }
  SCN_CHECK_HIP(hipMemcpyAsync(max_active_host, scratch_dev, 4, hipMemcpyDeviceToHost, stream));
  SCN_CHECK_HIP(hipStreamSynchronize(stream));
  return 0;
}

int scn_input_rules_table(const int32_t *order, const int32_t *site_start, int64_t N,
                          int32_t max_active, int32_t *table, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  if (N <= 0) return 0;
  k_input_table<<<grid_for(N * (1 + max_active)), 256, 0, stream>>>(order, site_start, N, max_active, table);
  SCN_CHECK_LAUNCH();
  return 0;
}

size_t scn_rules_temp_bytes(int64_t N, int K) {
  const int64_t nblk = (N + 255) / 256;
  return align256((size_t)K * (size_t)(nblk > 0 ? nblk : 1) * 4) + 256;
}

int scn_submanifold_rules(const int32_t *site_coords, int64_t N, const uint64_t *table_keys,
                          const int32_t *table_vals, int64_t capacity, int32_t *nbr,
                          int32_t *rules, int32_t *counts, void *temp, size_t temp_bytes,
                          const int32_t *site_ori, int mode, int only_class, int32_t *nbr_tmp, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(temp_bytes >= scn_rules_temp_bytes(N, 27), "scn_submanifold_rules: workspace too small");
  SCN_REQUIRE(mode == 0 || site_ori != nullptr, "scn_submanifold_rules: site_ori required for modes 1 and 2");
  SCN_REQUIRE(mode != 1 || nbr_tmp != nullptr, "scn_submanifold_rules: nbr_tmp [27][N] required for mode 1");
  SCN_REQUIRE(mode != 2 || only_class == 0 || only_class == 2 || only_class == 4, "scn_submanifold_rules: class must be 0, 2 or 4");
  if (N <= 0) {
    SCN_CHECK_HIP(hipMemsetAsync(counts, 0, 27 * 4, stream));
    return 0;
  }
  const int nblk = (int)((N + 255) / 256);
  int *blk_cnt = (int *)temp;
  k_sub_probe<<<nblk, 256, 0, stream>>>((const int4 *)site_coords, N, table_keys, table_vals,
                                       (uint32_t)(capacity - 1), mode == 1 ? nbr_tmp : nbr, blk_cnt, nblk, site_ori,
                                       mode == 2 ? only_class : -1);
  SCN_CHECK_LAUNCH();
  if (mode == 1) {
    k_nbr_remap<<<nblk, 256, 0, stream>>>(nbr_tmp, nbr, site_ori, N, blk_cnt, nblk);
    SCN_CHECK_LAUNCH();
  }
  k_excl_scan_rows<<<27, 1024, 0, stream>>>(blk_cnt, nblk, counts);
  SCN_CHECK_LAUNCH();
  k_rules_fill<27><<<nblk, 256, 0, stream>>>(nbr, N, blk_cnt, nblk, (int2 *)rules);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_strided_rules(const int32_t *fine_coords, int64_t N, const int32_t *coarse_of_fine,
                      int32_t *rules, int32_t *counts, void *temp, size_t temp_bytes,
                      void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(temp_bytes >= scn_rules_temp_bytes(N, 8), "scn_strided_rules: workspace too small");
  if (N <= 0) {
    SCN_CHECK_HIP(hipMemsetAsync(counts, 0, 8 * 4, stream));
    return 0;
  }
  const int nblk = (int)((N + 255) / 256);
  int *blk_cnt = (int *)temp;
  k_str_count<<<nblk, 256, 0, stream>>>((const int4 *)fine_coords, N, blk_cnt, nblk);
  SCN_CHECK_LAUNCH();
  k_excl_scan_rows<<<8, 1024, 0, stream>>>(blk_cnt, nblk, counts);
  SCN_CHECK_LAUNCH();
  k_str_fill<<<nblk, 256, 0, stream>>>((const int4 *)fine_coords, N, coarse_of_fine, blk_cnt, nblk,
                                      (int2 *)rules);
  SCN_CHECK_LAUNCH();
  return 0;
}

int scn_tile_passes(const int32_t *bases, int K, int64_t n_out, int32_t *passes, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  const int64_t n_gran = (n_out + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE, n_tiles = (n_out + 127) / 128;
  SCN_REQUIRE(bases != nullptr && passes != nullptr && K >= 1 && K <= 27 && n_out > 0 && n_tiles < (1 << 24),
              "scn_tile_passes: bad arguments");
  k_tile_passes<<<1, 1024, 0, stream>>>(bases, K, (int)n_gran, (int)n_tiles, passes);
  SCN_CHECK_LAUNCH();
  // (8-byte aligned behind the n_tiles + 1 prefix entries)
  k_sk_slices<<<dim3((SCN_SK_SLICES + 256) / 256, SCN_SK_MAX_NCB), 256, 0, stream>>>(passes, (int)n_tiles, (int2 *)(passes + ((n_tiles + 2) & ~(int64_t)1)));
  SCN_CHECK_LAUNCH();
  k_sk_slices_static<<<dim3(1, 3), 1024, 0, stream>>>(passes, (int)n_tiles, (int2 *)(passes + ((n_tiles + 2) & ~(int64_t)1)));
  SCN_CHECK_LAUNCH();
  return 0;
}

size_t scn_tile_passes_bytes(int64_t n_out) {
  const int64_t n_tiles = (n_out + 127) / 128;
  return (size_t)(((n_tiles + 2) & ~(int64_t)1) * 4 + (int64_t)SCN_SK_MAX_NCB * (SCN_SK_SLICES + 1) * 8);
}

int scn_tile_bases(const int32_t *rules, const int32_t *counts, int K, int64_t stride,
                   int out_col, int64_t n_out, int32_t *bases, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(out_col == 0 || out_col == 1, "scn_tile_bases: out_col must be 0 or 1");
  const int n_gran = (int)((n_out + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
  const int64_t total = (int64_t)K * (n_gran + 1);
  k_tile_bases<<<(int)((total + 255) / 256), 256, 0, stream>>>(rules, counts, K, stride, out_col,
                                                               n_gran, bases);
  SCN_CHECK_LAUNCH();
  return 0;
}

// ---- one call per batch (scn_batch_build): the launches of every grid and rulebook of a UNet's forward pass, into one arena
static int batch_layout(const scn_batch_plan *pl, scn_batch_layout *lo) {
  SCN_REQUIRE(pl != nullptr && lo != nullptr, "scn_batch_layout_query: NULL argument");
  const int L = pl->levels;
  SCN_REQUIRE(L >= 1 && L <= SCN_BATCH_MAX_LEVELS, "scn_batch: levels must be 1..SCN_BATCH_MAX_LEVELS");
  SCN_REQUIRE(pl->P > 0 && pl->P < (int64_t)1 << 31, "scn_batch: P out of range");
  SCN_REQUIRE(pl->ncols == 3 || pl->ncols == 4, "scn_batch: ncols must be 3 or 4");
  SCN_REQUIRE(pl->coords_dtype >= 0 && pl->coords_dtype <= 2, "scn_batch: coords_dtype must be 0 (int32), 1 (int64) or 2 (float32)");
  for (int l = 0; l < L; ++l) {
    SCN_REQUIRE(pl->N[l] > 0 && pl->N[l] <= (l == 0 ? pl->P : pl->N[l - 1]), "scn_batch: site counts must be positive and non-increasing");
    SCN_REQUIRE(!pl->strided[l] || l + 1 < L, "scn_batch: a strided rulebook needs the next scale");
  }
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return (int64_t)o; };
  for (int l = 0; l < SCN_BATCH_MAX_LEVELS; ++l)
    lo->coords[l] = lo->keys[l] = lo->hash_keys[l] = lo->hash_vals[l] = lo->sub_rules[l] = lo->sub_bases[l] = lo->str_rules[l] =
        lo->str_bases_fine[l] = lo->str_bases_coarse[l] = -1, lo->hash_cap[l] = 0;
  lo->small = take((size_t)L * (2 + 27 + 8) * 4);
  lo->order = take((size_t)pl->P * 4);
  lo->site_of_row = take((size_t)pl->P * 4);
  lo->site_start = take((size_t)(pl->P + 1) * 4);
  size_t sc_grid = 0, sc_rules = 0;  // scratch: one region for a grid build, one for a rulebook build, each the largest needed
  const bool occ = (pl->flags & 1) != 0;  // scn_batch_build_occ: no sort, no hash maps
  if (occ)  // (the input layer's arrival numbers, per-site counts and their block sums)
    sc_grid = align256((size_t)pl->P * 4) + align256((size_t)(pl->N[0] + 1) * 4) + align256((size_t)((pl->N[0] + 1 + 255) / 256 + 1) * 4);
  for (int l = 0; l < L; ++l) {
    const int64_t N = pl->N[l], Pl = l == 0 ? pl->P : pl->N[l - 1];
    const int64_t n_gran = (N + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE;
    lo->coords[l] = take((size_t)N * 16);
    lo->keys[l] = take((size_t)N * 8);
    if (!occ) {
      const size_t gt = scn_grid_build_temp_bytes(Pl);
      SCN_REQUIRE(gt > 0, "scn_batch: rocprim size query failed");
      // (grid scratch: the sort's workspace, the sorted keys, and for the coarser scales order / site_of_row / site_start)
      const size_t g = align256(gt) + align256((size_t)Pl * 8) + (l == 0 ? 0 : 2 * align256((size_t)Pl * 4) + align256((size_t)(Pl + 1) * 4));
      if (g > sc_grid) sc_grid = g;
    }
    if (pl->sub[l]) {
      if (!occ) {
        lo->hash_cap[l] = scn_hash_capacity(N);
        lo->hash_keys[l] = take((size_t)lo->hash_cap[l] * 8);
        lo->hash_vals[l] = take((size_t)lo->hash_cap[l] * 4);
      }
      lo->sub_rules[l] = take((size_t)27 * N * 8);
      lo->sub_bases[l] = take((size_t)27 * (n_gran + 1) * 4);
      const size_t r = align256((size_t)27 * N * 4) + align256(scn_rules_temp_bytes(N, 27));
      if (occ) sc_rules += r;  // (every scale's neighbour table and block counts at once: one launch for all scales)
      else if (r > sc_rules) sc_rules = r;
    }
    if (pl->strided[l]) {
      const int64_t n_gran_c = (pl->N[l + 1] + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE;
      lo->str_rules[l] = take((size_t)8 * N * 8);
      lo->str_bases_fine[l] = take((size_t)8 * (n_gran + 1) * 4);
      lo->str_bases_coarse[l] = take((size_t)8 * (n_gran_c + 1) * 4);
      const size_t r = align256(scn_rules_temp_bytes(N, 8));
      if (occ) sc_rules += r;
      else if (r > sc_rules) sc_rules = r;
    }
  }
  lo->scratch = (int64_t)off;
  off += sc_grid + sc_rules;
  lo->total = (int64_t)off;
  return 0;
}

int scn_batch_layout_query(const scn_batch_plan *plan_host, scn_batch_layout *layout_host) { return batch_layout(plan_host, layout_host); }

int scn_batch_build(const void *coords, const scn_batch_plan *pl, void *arena, size_t arena_bytes, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  scn_batch_layout lo;
  SCN_REQUIRE(pl != nullptr && !(pl->flags & 1), "scn_batch_build: a plan with flags bit 0 is for scn_batch_build_occ");
  if (int rc = batch_layout(pl, &lo)) return rc;
  SCN_REQUIRE(coords != nullptr && arena != nullptr && arena_bytes >= (size_t)lo.total, "scn_batch_build: arena smaller than scn_batch_layout_query says");
  char *A = (char *)arena;
  const int L = pl->levels;
  int32_t *small = (int32_t *)(A + lo.small);
  // grid scratch, the same for every scale (the stream runs them one after the other)
  size_t sc_grid = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t Pl = l == 0 ? pl->P : pl->N[l - 1];
    const size_t g = align256(scn_grid_build_temp_bytes(Pl)) + align256((size_t)Pl * 8) +
                     (l == 0 ? 0 : 2 * align256((size_t)Pl * 4) + align256((size_t)(Pl + 1) * 4));
    if (g > sc_grid) sc_grid = g;
  }
  char *SG = A + lo.scratch, *SR = SG + sc_grid;
  for (int l = 0; l < L; ++l) {
    const int64_t N = pl->N[l], Pl = l == 0 ? pl->P : pl->N[l - 1];
    // ---- the grid of scale l: from the input coordinates (l = 0) or from the site coordinates of scale l - 1, shifted by one bit
    const size_t gt = scn_grid_build_temp_bytes(Pl);
    char *q = SG;
    void *temp = q; q += align256(gt);
    uint64_t *keys_sorted = (uint64_t *)q; q += align256((size_t)Pl * 8);
    int32_t *order, *site_of_row, *site_start;
    if (l == 0) {
      order = (int32_t *)(A + lo.order); site_of_row = (int32_t *)(A + lo.site_of_row); site_start = (int32_t *)(A + lo.site_start);
    } else {
      order = (int32_t *)q; q += align256((size_t)Pl * 4);
      site_of_row = (int32_t *)q; q += align256((size_t)Pl * 4);
      site_start = (int32_t *)q;
    }
    const void *src = l == 0 ? coords : (const void *)(A + lo.coords[l - 1]);
    const int dt = l == 0 ? pl->coords_dtype : 0, nc = l == 0 ? pl->ncols : 4, shift = l == 0 ? 0 : 1;
    if (int rc = grid_build_launches(src, dt, nc, shift, Pl, temp, gt, keys_sorted, order, site_of_row, site_start, small + 2 * l, stream))
      return rc;
    if (int rc = scn_grid_finalize(src, dt, nc, shift, Pl, N, keys_sorted, order, site_start, (int32_t *)(A + lo.coords[l]),
                                   (uint64_t *)(A + lo.keys[l]), stream_))
      return rc;
    // ---- the 2/2 rulebook scale l - 1 -> l (site_of_row of this grid build = the coarse site of every finer site)
    if (l > 0 && pl->strided[l - 1]) {
      int32_t *rules = (int32_t *)(A + lo.str_rules[l - 1]);
      int32_t *counts = small + 2 * L + 27 * L + 8 * (l - 1);
      if (int rc = scn_strided_rules((const int32_t *)(A + lo.coords[l - 1]), Pl, site_of_row, rules, counts, SR, scn_rules_temp_bytes(Pl, 8), stream_))
        return rc;
      if (int rc = scn_tile_bases(rules, counts, 8, Pl, 1, N, (int32_t *)(A + lo.str_bases_coarse[l - 1]), stream_)) return rc;
      if (int rc = scn_tile_bases(rules, counts, 8, Pl, 0, Pl, (int32_t *)(A + lo.str_bases_fine[l - 1]), stream_)) return rc;
    }
    // ---- hash map and submanifold rulebook of scale l
    if (pl->sub[l]) {
      uint64_t *hk = (uint64_t *)(A + lo.hash_keys[l]);
      int32_t *hv = (int32_t *)(A + lo.hash_vals[l]);
      if (int rc = scn_hash_build((const uint64_t *)(A + lo.keys[l]), N, hk, hv, lo.hash_cap[l], stream_)) return rc;
      int32_t *nbr = (int32_t *)SR;
      void *rt = SR + align256((size_t)27 * N * 4);
      int32_t *rules = (int32_t *)(A + lo.sub_rules[l]);
      int32_t *counts = small + 2 * L + 27 * l;
      if (int rc = scn_submanifold_rules((const int32_t *)(A + lo.coords[l]), N, hk, hv, lo.hash_cap[l], nbr, rules, counts, rt,
                                         scn_rules_temp_bytes(N, 27), nullptr, 0, -1, nullptr, stream_))
        return rc;
      if (int rc = scn_tile_bases(rules, counts, 27, N, 1, N, (int32_t *)(A + lo.sub_bases[l]), stream_)) return rc;
    }
  }
  return 0;
}

// ---- occupancy bitmaps (include/scn_hip.h) ----
int scn_occ_bounds(const void *coords, int coords_dtype, int ncols, int64_t P, int32_t *scratch_dev, int32_t *minmax_host,
                   void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(P > 0 && P < (int64_t)1 << 31, "scn_occ_bounds: P out of range");
  SCN_REQUIRE(ncols == 3 || ncols == 4, "scn_occ_bounds: ncols must be 3 or 4");
  SCN_REQUIRE(coords_dtype >= 0 && coords_dtype <= 2, "scn_occ_bounds: coords_dtype must be 0 (int32), 1 (int64) or 2 (float32)");
  SCN_REQUIRE(scratch_dev != nullptr && minmax_host != nullptr, "scn_occ_bounds: NULL argument");
  SCN_CHECK_HIP(hipMemsetAsync(scratch_dev, 0x80, 36, stream));
  const int g = grid_for(P, 256, 512);
  if (coords_dtype == 1) k_occ_bounds<1><<<g, 256, 0, stream>>>(coords, ncols, P, scratch_dev);
  else if (coords_dtype == 0) k_occ_bounds<0><<<g, 256, 0, stream>>>(coords, ncols, P, scratch_dev);
  else k_occ_bounds<2><<<g, 256, 0, stream>>>(coords, ncols, P, scratch_dev);
  SCN_CHECK_LAUNCH();
  int32_t h[9];
  SCN_CHECK_HIP(hipMemcpyAsync(h, scratch_dev, 36, hipMemcpyDeviceToHost, stream));
  SCN_CHECK_HIP(hipStreamSynchronize(stream));
  if (h[8] != (int32_t)0x80808080 || h[4] == (int32_t)0x80808080)
    return fail(3, "scn_occ_bounds: coordinate outside the 31-bit key box (0<=x<1024, 0<=y<2048, 0<=z<1024) or bad sample index%s", "");
  for (int a = 0; a < 4; ++a) {
    minmax_host[a] = -h[a];
    minmax_host[4 + a] = h[4 + a];
  }
  return 0;
}

int scn_occ_plan(const int32_t *mm, int levels, scn_occ_box *box) {
  SCN_REQUIRE(mm != nullptr && box != nullptr, "scn_occ_plan: NULL argument");
  SCN_REQUIRE(levels >= 1 && levels <= SCN_BATCH_MAX_LEVELS, "scn_occ_plan: levels must be 1..SCN_BATCH_MAX_LEVELS");
  for (int a = 0; a < 4; ++a) SCN_REQUIRE(mm[a] >= 0 && mm[4 + a] >= mm[a], "scn_occ_plan: bad bounds");
  memset(box, 0, sizeof(*box));
  box->levels = levels;
  const int align = 1 << (levels - 1);
  for (int a = 0; a < 3; ++a) box->origin[a] = mm[a] & ~(align - 1);
  box->origin[3] = mm[3];
  box->nb = mm[7] - mm[3] + 1;
  int64_t W = 0;
  for (int l = 0; l < levels; ++l) {
    for (int a = 0; a < 3; ++a) box->n[l][a] = ((mm[4 + a] - box->origin[a]) >> l) + 1;
    box->wx[l] = (box->n[l][0] + 31) / 32;
    box->woff[l] = W;
    W += (int64_t)box->nb * box->n[l][2] * box->n[l][1] * box->wx[l];
    if (W > SCN_OCC_MAX_WORDS) {
      box->bytes = 0;  // (too large: the caller builds with scn_batch_build)
      return 0;
    }
  }
  box->woff[levels] = W;
  box->blocks = (W + 1023) / 1024;
  box->off_info = OCC_HDR_BYTES;
  box->off_blk = OCC_HDR_BYTES + 1024;
  box->off_words = (int64_t)align256((size_t)box->off_blk + (size_t)(box->blocks + 1) * 4);
  box->bytes = box->off_words + box->blocks * 1024 * 8;
  return 0;
}

static OccDev occ_dev(const scn_occ_box *box) {
  OccDev d;
  memset(&d, 0, sizeof(d));
  d.levels = box->levels; d.nb = box->nb;
  d.ox = box->origin[0]; d.oy = box->origin[1]; d.oz = box->origin[2]; d.ob = box->origin[3];
  for (int l = 0; l < box->levels; ++l) {
    d.nx[l] = box->n[l][0]; d.ny[l] = box->n[l][1]; d.nz[l] = box->n[l][2]; d.wx[l] = box->wx[l];
  }
  for (int l = 0; l <= box->levels; ++l) d.woff[l] = box->woff[l];
  return d;
}

int scn_occ_chain(const void *coords, int coords_dtype, int ncols, int64_t P, const scn_occ_box *box, void *occ, size_t occ_bytes,
                  int32_t *n_sites_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(box != nullptr && box->bytes > 0 && occ != nullptr && occ_bytes >= (size_t)box->bytes, "scn_occ_chain: buffer smaller than scn_occ_plan says");
  SCN_REQUIRE(P > 0 && P < (int64_t)1 << 31, "scn_occ_chain: P out of range");
  SCN_REQUIRE(ncols == 3 || ncols == 4, "scn_occ_chain: ncols must be 3 or 4");
  SCN_REQUIRE(coords_dtype >= 0 && coords_dtype <= 2, "scn_occ_chain: coords_dtype must be 0 (int32), 1 (int64) or 2 (float32)");
  char *B = (char *)occ;
  uint64_t *buf = (uint64_t *)B, *words = (uint64_t *)(B + box->off_words);
  int *info = (int *)(B + box->off_info), *blk = (int *)(B + box->off_blk);
  // (the counters and everything behind them: the block sums are written before they are read, but one memset is one launch)
  SCN_CHECK_HIP(hipMemsetAsync(B + box->off_info, 0, (size_t)(box->bytes - box->off_info), stream));
  k_put<OccDev><<<1, 64, 0, stream>>>(occ_dev(box), (OccDev *)B);
  SCN_CHECK_LAUNCH();
  const int nblk = (int)((P + 255) / 256);
  if (coords_dtype == 1) k_occ_set<1><<<nblk, 256, 0, stream>>>(coords, ncols, P, buf, words);
  else if (coords_dtype == 0) k_occ_set<0><<<nblk, 256, 0, stream>>>(coords, ncols, P, buf, words);
  else k_occ_set<2><<<nblk, 256, 0, stream>>>(coords, ncols, P, buf, words);
  SCN_CHECK_LAUNCH();
  for (int l = 0; l + 1 < box->levels; ++l) {
    const int64_t Wc = box->woff[l + 2] - box->woff[l + 1];
    k_occ_coarsen<<<(int)((Wc + 255) / 256), 256, 0, stream>>>(buf, words, l);
    SCN_CHECK_LAUNCH();
  }
  k_occ_count<<<(int)box->blocks, 256, 0, stream>>>(words, blk);
  SCN_CHECK_LAUNCH();
  k_excl_scan_rows<<<1, 1024, 0, stream>>>(blk, (int)box->blocks, info + OCC_INFO_START + box->levels);
  SCN_CHECK_LAUNCH();
  k_occ_base<<<(int)box->blocks, 256, 0, stream>>>(buf, words, blk, info);
  SCN_CHECK_LAUNCH();
  if (n_sites_host == nullptr) return 0;
  return scn_occ_fetch(occ, box, n_sites_host, stream_);
}

int scn_occ_fetch(const void *occ, const scn_occ_box *box, int32_t *n_sites_host, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  SCN_REQUIRE(occ != nullptr && box != nullptr && box->bytes > 0 && n_sites_host != nullptr, "scn_occ_fetch: bad arguments");
  int h[SCN_BATCH_MAX_LEVELS + 1];
  SCN_CHECK_HIP(hipMemcpyAsync(h, (const char *)occ + box->off_info, 4 * (box->levels + 1), hipMemcpyDeviceToHost, stream));
  SCN_CHECK_HIP(hipStreamSynchronize(stream));
  for (int l = 0; l < box->levels; ++l) n_sites_host[l] = h[l + 1] - h[l];
  return 0;
}

int scn_batch_build_occ(const void *coords, const scn_batch_plan *pl, const scn_occ_box *box, void *occ, void *arena,
                        size_t arena_bytes, void *scratch, size_t scratch_bytes, void *stream_) {
  hipStream_t stream = as_stream(stream_);
  scn_batch_layout lo;
  SCN_REQUIRE(pl != nullptr && (pl->flags & 1), "scn_batch_build_occ: the plan's flags bit 0 must be set");
  if (int rc = batch_layout(pl, &lo)) return rc;
  // the scratch behind the results (one allocation of layout.total bytes), or an allocation of its own that the caller drops
  // after the call while the results live as long as the batch
  SCN_REQUIRE(coords != nullptr && arena != nullptr && arena_bytes >= (size_t)(scratch != nullptr ? lo.scratch : lo.total),
              "scn_batch_build_occ: arena smaller than scn_batch_layout_query says");
  SCN_REQUIRE(scratch == nullptr || scratch_bytes >= (size_t)(lo.total - lo.scratch), "scn_batch_build_occ: scratch smaller than layout.total - layout.scratch");
  SCN_REQUIRE(box != nullptr && box->bytes > 0 && occ != nullptr && box->levels >= pl->levels, "scn_batch_build_occ: the box must cover the plan's scales");
  char *A = (char *)arena, *B = (char *)occ;
  const uint64_t *buf = (const uint64_t *)B, *words = (const uint64_t *)(B + box->off_words);
  int *info = (int *)(B + box->off_info);
  const int L = pl->levels;
  const int64_t P = pl->P, N0 = pl->N[0];
  int32_t *small = (int32_t *)(A + lo.small);
  // ---- what the multi-scale launches read: sizes, block ranges, where everything lies
  char *S = scratch != nullptr ? (char *)scratch : A + lo.scratch, *const S0 = S;  // (offsets below are relative to the arena either way)
  int32_t *slot = (int32_t *)S; S += align256((size_t)P * 4);
  int *cnt = (int *)S; S += align256((size_t)(N0 + 1) * 4);
  const int nb0 = (int)((N0 + 1 + 255) / 256);
  int *blk = (int *)S; S += align256((size_t)(nb0 + 1) * 4);
  BatchDev bd;
  TileDescs td;
  memset(&bd, 0, sizeof(bd));
  memset(&td, 0, sizeof(td));
  bd.L = L;
  bd.small = lo.small;
  int64_t nblocks = 0, threads = 0;
  bool any_sub = false, any_str = false;
  auto tile = [&](int64_t rules, int64_t bases, int counts, int K, int64_t stride, int out_col, int64_t n_out) {
    TileDesc &q = td.d[td.n++];
    q.rules = rules; q.bases = bases; q.stride = stride; q.first = threads;
    q.counts = counts; q.K = K; q.out_col = out_col;
    q.n_gran = (int)((n_out + SCN_TILE_GRANULE - 1) / SCN_TILE_GRANULE);
    threads += (int64_t)K * (q.n_gran + 1);
  };
  for (int l = 0; l < SCN_BATCH_MAX_LEVELS; ++l) bd.sub_counts[l] = bd.str_counts[l] = -1;
  for (int l = 0; l < L; ++l) {
    const int64_t N = pl->N[l];
    const int64_t nblk = (N + 255) / 256;
    bd.N[l] = (int32_t)N;
    bd.blk0[l] = (int32_t)nblocks;
    nblocks += nblk;
    bd.coords[l] = lo.coords[l];
    bd.keys[l] = lo.keys[l];
    if (pl->sub[l]) {
      any_sub = true;
      bd.sub_counts[l] = 2 * L + 27 * l;
      bd.sub_rules[l] = lo.sub_rules[l];
      bd.nbr[l] = S - A; S += align256((size_t)27 * N * 4);
      bd.sub_blk[l] = S - A; S += align256((size_t)27 * nblk * 4);
      bd.sub_bases[l] = lo.sub_bases[l];  // (written by the fill launch)
    }
    if (pl->strided[l]) {
      any_str = true;
      bd.str_counts[l] = 2 * L + 27 * L + 8 * l;
      bd.str_rules[l] = lo.str_rules[l];
      bd.str_blk[l] = S - A; S += align256((size_t)8 * nblk * 4);
      bd.str_bases_fine[l] = lo.str_bases_fine[l];  // (written by the fill launch; the coarse column's needs the search)
      tile(lo.str_rules[l], lo.str_bases_coarse[l], bd.str_counts[l], 8, N, 1, pl->N[l + 1]);
    }
  }
  for (int l = L; l <= SCN_BATCH_MAX_LEVELS; ++l) bd.blk0[l] = (int32_t)nblocks;
  td.threads = threads;
  SCN_REQUIRE((size_t)(S - S0) <= (size_t)(lo.total - lo.scratch) && nblocks < (int64_t)1 << 31, "scn_batch_build_occ: scratch layout");
  k_put<BatchDev><<<1, 64, 0, stream>>>(bd, (BatchDev *)(B + OCC_HDR_BATCH));
  SCN_CHECK_LAUNCH();
  k_put<TileDescs><<<1, 64, 0, stream>>>(td, (TileDescs *)(B + OCC_HDR_TILES));
  SCN_CHECK_LAUNCH();
  // ---- sites of every scale
  const int64_t W = box->woff[box->levels];
  k_occ_sites<<<(int)((W + 255) / 256), 256, 0, stream>>>(buf, words, W, info, A);
  SCN_CHECK_LAUNCH();
  // ---- input layer: site_of_row, site_start, order
  int32_t *site_of_row = (int32_t *)(A + lo.site_of_row), *site_start = (int32_t *)(A + lo.site_start), *order = (int32_t *)(A + lo.order);
  SCN_CHECK_HIP(hipMemsetAsync(cnt, 0, (size_t)(N0 + 1) * 4, stream));
  const int nbp = (int)((P + 255) / 256);
  if (pl->coords_dtype == 1) k_occ_rows<1><<<nbp, 256, 0, stream>>>(coords, pl->ncols, P, buf, words, info, site_of_row, slot, cnt);
  else if (pl->coords_dtype == 0) k_occ_rows<0><<<nbp, 256, 0, stream>>>(coords, pl->ncols, P, buf, words, info, site_of_row, slot, cnt);
  else k_occ_rows<2><<<nbp, 256, 0, stream>>>(coords, pl->ncols, P, buf, words, info, site_of_row, slot, cnt);
  SCN_CHECK_LAUNCH();
  k_sum256<<<nb0, 256, 0, stream>>>(cnt, N0, blk);
  SCN_CHECK_LAUNCH();
  k_excl_scan_rows<<<1, 1024, 0, stream>>>(blk, nb0, blk + nb0);
  SCN_CHECK_LAUNCH();
  k_starts256<<<nb0, 256, 0, stream>>>(cnt, N0, blk, site_start, (int)P);
  SCN_CHECK_LAUNCH();
  k_occ_order<<<nbp, 256, 0, stream>>>(site_of_row, slot, site_start, P, order);
  SCN_CHECK_LAUNCH();
  k_occ_order_sort<<<(int)((N0 + 255) / 256), 256, 0, stream>>>(site_start, N0, order, info);
  SCN_CHECK_LAUNCH();
  k_occ_small<<<1, 64, 0, stream>>>(info, L, small);
  SCN_CHECK_LAUNCH();
  // ---- rulebooks of every scale: six launches (the fills write the tile-base tables of their sorted column themselves)
  if (any_sub) {
    k_sub_probe_occ<<<(int)nblocks, 256, 0, stream>>>(buf, words, info, A);
    SCN_CHECK_LAUNCH();
  }
  if (any_str) {
    k_str_count_ml<<<(int)nblocks, 256, 0, stream>>>(buf, A);
    SCN_CHECK_LAUNCH();
  }
  k_scan_ml<<<L * 35, 1024, 0, stream>>>(buf, A);
  SCN_CHECK_LAUNCH();
  if (any_sub) {
    k_rules_fill_ml<<<(int)nblocks, 256, 0, stream>>>(buf, A);
    SCN_CHECK_LAUNCH();
  }
  if (any_str) {
    k_str_fill_ml<<<(int)nblocks, 256, 0, stream>>>(buf, words, info, A);
    SCN_CHECK_LAUNCH();
  }
  if (threads > 0) {
    k_tile_bases_ml<<<(int)((threads + 255) / 256), 256, 0, stream>>>(buf, A);
    SCN_CHECK_LAUNCH();
  }
  return 0;
}

}  // extern "C"
