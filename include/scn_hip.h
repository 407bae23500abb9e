/*
 * scn_hip.h -- C ABI of libscn_hip.so: the MI355X (gfx950) sparse-convolution
 * hot path behind the reference's `sparseconvnet.SCN` extension boundary.
 *
 * Every entry point replaces one (or the device half of one) function the
 * reference binds in sparseconvnet/SCN/pybind.cpp and dispatches in
 * sparseconvnet/SCN/sparseconvnet_cuda.cpp.  Paths below are relative to the
 * reference tree.  Conventions:
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - features are contiguous fp32 [rows, planes]; indices are int32 (`Int`,
 *     sparseconvnet/SCN/Metadata/32bits.h:11);
 *   - `stream` is a hipStream_t passed as void*; nothing runs on another stream;
 *   - the only entry points that synchronise the stream are those whose
 *     result sizes the next allocation (scn_grid_build, scn_input_rules_max_active,
 *     scn_grid_chain_counts / _fetch, scn_occ_bounds, scn_occ_chain / scn_occ_fetch);
 *   - no entry point allocates device memory: callers pass workspaces whose
 *     sizes are given by the *_bytes query functions;
 *   - return value 0 = ok, otherwise an error code; scn_last_error() holds text.
 *
 * Rulebook layout ("rules"): int32 [K][stride][2]; list k holds counts[k] pairs
 * (col0, col1) exactly as the reference's RuleBook rules[k] = {in0,out0,in1,out1,..}
 * (sparseconvnet/SCN/Metadata/Metadata.h:73), bit for bit, but device resident.
 */
#ifndef SCN_HIP_H
#define SCN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCN_ABI_VERSION 7
#define SCN_TILE_GRANULE 32 /* output-row granule of the tile-base tables */
#define SCN_SK_STATE_BYTES 65536 /* scn_conv_forward's sk_state */
#define SCN_GRID_CHAIN_MAX 12     /* scales one scn_grid_chain_counts call can count */
#define SCN_DW_STATE_BYTES 1048576 /* scn_conv_dweight's dw_state */

int scn_abi_version(void);
const char *scn_last_error(void);

/* ---------------------------------------------------------------- grids ---
 * Voxel grid of one scale = the reference's GPU_SparseGrids + Multival_Point_Hashtable
 * (sparseconvnet/SCN/Metadata/Metadata.h:114-120, sparseconvnet/SCN/CUDA/CUDPPWrapper.hpp:731-999).
 */

/* workspace bytes needed by scn_grid_build for P input rows */
size_t scn_grid_build_temp_bytes(int64_t P);

/* Number the active sites.  Replaces HashInputPointClouds + InsertAndCompactPointCloud
 * (sparseconvnet/SCN/CUDA/CUDPPWrapper.hpp:663-681,:767-793; key :CUDPPWrapper.cu:80-83; rank
 * semantics extra/cudpp/src/cudpp_hash/hash_multivalue.cpp:45-93) for the input
 * layer (coords int64 [P,ncols], ncols 3|4, shift 0) and the `insert` of the
 * coarse grid in Convolution_InputSgToRulesAndOutputSg_FastDownSampleMode
 * (sparseconvnet/SCN/Metadata/ConvolutionRules.h:370-372; coords = fine site coords int32 [P,4], shift 1).
 *   keys_sorted [P] u64   (b<<31 | key31) ascending
 *   order       [P] i32   original row of each sorted position (stable)
 *   site_of_row [P] i32   global site id of each input row  (`_d_hash_idx`)
 *   site_start  [P+1] i32 first sorted position of each site; site_start[N] = P
 *   n_sites_host          N, written after ONE stream synchronisation
 * coords_dtype: 0 = int32, 1 = int64, 2 = float32 [P,ncols] as the reference's collate functions produce them
 * (examples/ScanNet/datasets/scannet.py:254-265), converted like sparseconvnet/ioLayers.py:56 does on the host
 * (`.type(torch.LongTensor)`: truncation toward zero) -- no int64 copy, no host round trip.
 * error if a coordinate lies outside 0<=x<1024, 0<=y<2048, 0<=z<1024 (the
 * reference silently aliases there, sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:80-81). */
int scn_grid_build(const void *coords, int coords_dtype, int ncols, int shift, int64_t P,
                   void *temp, size_t temp_bytes, uint64_t *keys_sorted, int32_t *order,
                   int32_t *site_of_row, int32_t *site_start, int32_t *n_sites_host,
                   void *stream);

/* One host synchronisation per BATCH instead of one per scale.  The reference builds the grid of every scale when the
 * first convolution that needs it asks (Metadata::getRuleBook -> Convolution_InputSgsToRulesAndOutputSgs_OMP /
 * ..._FastDownSampleMode, sparseconvnet/SCN/Metadata/ConvolutionRules.h:741-843, :353-372) and learns its size there;
 * all of them are functions of the input coordinates alone (scale l holds the voxels of coords >> l for size = stride =
 * 2).  scn_grid_chain_counts counts the distinct voxels of coords >> l for l = 0..levels-1 in one launch (hash sets in
 * `temp`) and reads the `levels` counts back with ONE stream synchronisation; scn_grid_build_async is scn_grid_build
 * without its read-back for a caller that already knows N: it leaves N and the error flag in status_dev[0..1]
 * (device memory) for the caller to compare with what it assumed at its next read-back.  Same errors as
 * scn_grid_build. */
size_t scn_grid_chain_temp_bytes(int64_t P, int levels);
int scn_grid_chain_counts(const void *coords, int coords_dtype, int ncols, int64_t P, int levels, void *temp,
                          size_t temp_bytes, int32_t *n_sites_host, void *stream);
/* n_sites_host = NULL above: the counting launch is only queued; this reads the counts back (the one synchronisation) -- a caller
 * queues the launches of the first grid, whose sizes depend on P alone, in between, so that the device works while the host waits */
int scn_grid_chain_fetch(const void *temp, int64_t P, int levels, int32_t *n_sites_host, void *stream);
int scn_grid_build_async(const void *coords, int coords_dtype, int ncols, int shift, int64_t P,
                         void *temp, size_t temp_bytes, uint64_t *keys_sorted, int32_t *order,
                         int32_t *site_of_row, int32_t *site_start, int32_t *status_dev, void *stream);

/* site_coords [N,4] i32 = (x,y,z,b) of the first duplicate of each site
 * (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:224-226), site_keys [N] u64. */
int scn_grid_finalize(const void *coords, int coords_dtype, int ncols, int shift, int64_t P,
                      int64_t N, const uint64_t *keys_sorted, const int32_t *order,
                      const int32_t *site_start, int32_t *site_coords, uint64_t *site_keys,
                      void *stream);

/* open-addressing voxel hash map key -> site id; capacity (power of two >= 2N) */
int64_t scn_hash_capacity(int64_t N);
int scn_hash_build(const uint64_t *site_keys, int64_t N, uint64_t *table_keys,
                   int32_t *table_vals, int64_t capacity, void *stream);

/* padded input rule table [N, 1+maxActive] in the reference's host format
 * (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:53-64; sparseconvnet/SCN/Metadata/IOLayersRules.h:181-199).
 * Not used by the hot path (which reads order/site_start directly). */
int scn_input_rules_max_active(const int32_t *site_start, int64_t N, int32_t *scratch_dev,
                               int32_t *max_active_host, void *stream);
int scn_input_rules_table(const int32_t *order, const int32_t *site_start, int64_t N,
                          int32_t max_active, int32_t *table, void *stream);

/* ------------------------------------------------------------ rulebooks --- */

/* 3x3x3 submanifold rulebook.  Replaces SubmanifoldBuildObject::{generateQueryList,
 * global_hash_query, generate_rulebook} (sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cpp:97-202;
 * kernels sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:45-114) and the 27 CUDPP compactions.
 *   nbr    [27][N] i32   neighbour site per offset or -1
 *   rules  [27][N][2] i32, counts [27] i32 (device)
 *   temp   scn_rules_temp_bytes(N, 27) bytes
 * Normal-guided filters (the fork's use_normal path): site_ori [N] = orientation class of each site (0, 2 or 4:
 * OrientedFilter, sparseconvnet/SCN/Metadata/RectangularRegions.h:12-32, see scn_site_normals).
 *   mode 0: plain rulebook (site_ori, nbr_tmp may be NULL);
 *   mode 1: remap_rules_with_normal (sparseconvnet/SCN/Metadata/SubmanifoldConvolutionRules.h:213-245): the pair found at
 *           offset i for output site o goes to list rot_index[ori(o)][i]; every list sorted by output site (the
 *           reference's lists hold the same pairs as up to three such runs, ordered by i); nbr_tmp [27][N] scratch;
 *   mode 2: plain offsets, but only the output sites of class `only_class` (the three such rulebooks are what the
 *           dInput of a remapped convolution runs over: in them the input column is unique and ascending too). */
size_t scn_rules_temp_bytes(int64_t N, int K);
int scn_submanifold_rules(const int32_t *site_coords, int64_t N, const uint64_t *table_keys,
                          const int32_t *table_vals, int64_t capacity, int32_t *nbr,
                          int32_t *rules, int32_t *counts, void *temp, size_t temp_bytes,
                          const int32_t *site_ori, int mode, int only_class, int32_t *nbr_tmp,
                          void *stream);

/* size 2 / stride 2 rulebook: pairs (fine, coarse), 8 lists, ascending fine.
 * Replaces d_Convolution_GenerateOutputRules
 * (sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:534-628).  rules [8][N][2]. */
int scn_strided_rules(const int32_t *fine_coords, int64_t N, const int32_t *coarse_of_fine,
                      int32_t *rules, int32_t *counts, void *temp, size_t temp_bytes,
                      void *stream);

/* bases [K][n_granules+1]: index of the first pair of list k whose column
 * `out_col` is >= g*SCN_TILE_GRANULE (lists are sorted on that column);
 * n_granules = ceil(n_out / SCN_TILE_GRANULE). */
int scn_tile_bases(const int32_t *rules, const int32_t *counts, int K, int64_t stride,
                   int out_col, int64_t n_out, int32_t *bases, void *stream);

/* passes (scn_tile_passes_bytes(n_out) bytes) = int32 [ceil(n_out / 128) + 1]: prefix sums over the 128-row output tiles of
 * max(1, sum_k ceil(pairs_k(tile) / 32)) -- the number of 32-pair passes the big gather-mm launches (k_conv_tile4) make over a
 * tile per 64-plane input chunk -- from the tile-base table of the output column; behind it the slice tables of a stream-K
 * launch (scn_conv_forward's `tile_passes`): where each of its 1024 slices begins, for 1 / 2 / 4 column blocks.  The reference's
 * chunk kernels (sparseconvnet/SCN/CUDA/Convolution.cu:1059-1152) deal whole chunks in launch order. */
size_t scn_tile_passes_bytes(int64_t n_out); /* bytes of `passes`: the prefix, then one slice table per column-block count */
int scn_tile_passes(const int32_t *bases, int K, int64_t n_out, int32_t *passes, void *stream);

/* ---------------------------------------------- all grids and rulebooks of a batch, one call ---
 * What a forward pass of the UNet asks the reference's Metadata for one piece at a time -- Metadata::inputLayer, then per scale
 * getSubmanifoldRuleBook and getRuleBook (sparseconvnet/SCN/Metadata/Metadata.cpp:424-446, :504-529, :598-625; drivers
 * sparseconvnet/SCN/Metadata/ConvolutionRules.h:741-843, sparseconvnet/SCN/Metadata/SubmanifoldConvolutionRules.h:248-275) -- queued
 * by ONE call into ONE caller-allocated arena, for a caller that knows the site count of every scale (scn_grid_chain_counts).
 * Same kernels, same arrays, bit for bit, as scn_grid_build_async / scn_grid_finalize / scn_hash_build / scn_submanifold_rules
 * / scn_strided_rules / scn_tile_bases called one by one: ~60 calls and ~120 allocations of the host per batch become two
 * (round 5: the forward-only configuration is bound by host time, a third of it the rule build).
 *   plan:   scale 0 = the input grid of `coords` [P, ncols]; scale l = scale l-1 coarsened by size = stride = 2.
 *   layout: byte offsets into the arena (256-byte aligned; -1 = not built).  `small` = int32 [levels][2] grid status (N found,
 *           error flag -- compare N with plan->N), then [levels][27] submanifold list lengths, then [levels][8] strided list
 *           lengths: ONE device-to-host copy of 4 * levels * 37 bytes hands the host everything it needs to read.
 *           Everything in front of `scratch` is a result; the scratch part may be freed when the stream has passed the call.
 * scn_batch_layout_query launches nothing (it asks rocPRIM for the sort's workspace size, which needs a device to be present). */
#define SCN_BATCH_MAX_LEVELS 12
typedef struct {
  int32_t levels;                          /* 1 .. SCN_BATCH_MAX_LEVELS */
  int32_t coords_dtype, ncols;             /* of the input coordinates: see scn_grid_build */
  int32_t flags;                           /* bit 0: layout for scn_batch_build_occ (no hash maps, no sort workspace) */
  int64_t P;                               /* input rows */
  int64_t N[SCN_BATCH_MAX_LEVELS];         /* active sites of every scale (> 0) */
  int32_t sub[SCN_BATCH_MAX_LEVELS];       /* != 0: hash map + 3x3x3 submanifold rulebook + tile bases of this scale */
  int32_t strided[SCN_BATCH_MAX_LEVELS];   /* != 0: 2/2 rulebook scale l -> l+1 + both tile-base tables (l + 1 < levels) */
} scn_batch_plan;
typedef struct {
  int64_t total, scratch, small;
  int64_t order, site_of_row, site_start;                   /* input layer: int32 [P], [P], [P+1] */
  int64_t coords[SCN_BATCH_MAX_LEVELS];                     /* int32 [N_l][4] */
  int64_t keys[SCN_BATCH_MAX_LEVELS];                       /* u64 [N_l] */
  int64_t hash_keys[SCN_BATCH_MAX_LEVELS];                  /* u64 [hash_cap_l] */
  int64_t hash_vals[SCN_BATCH_MAX_LEVELS];                  /* int32 [hash_cap_l] */
  int64_t hash_cap[SCN_BATCH_MAX_LEVELS];                   /* (a count, not an offset) */
  int64_t sub_rules[SCN_BATCH_MAX_LEVELS];                  /* int32 [27][N_l][2] */
  int64_t sub_bases[SCN_BATCH_MAX_LEVELS];                  /* int32 [27][ceil(N_l / 32) + 1] */
  int64_t str_rules[SCN_BATCH_MAX_LEVELS];                  /* int32 [8][N_l][2]: pairs (site of scale l, site of scale l+1) */
  int64_t str_bases_fine[SCN_BATCH_MAX_LEVELS];             /* int32 [8][ceil(N_l / 32) + 1]      (out_col 0) */
  int64_t str_bases_coarse[SCN_BATCH_MAX_LEVELS];           /* int32 [8][ceil(N_{l+1} / 32) + 1]  (out_col 1) */
} scn_batch_layout;
int scn_batch_layout_query(const scn_batch_plan *plan_host, scn_batch_layout *layout_host);
int scn_batch_build(const void *coords, const scn_batch_plan *plan_host, void *arena, size_t arena_bytes, void *stream);

/* ------------------------------- the same batch from OCCUPANCY BITMAPS: no sort, no hash map (round 6) ---
 * The reference numbers the sites of a grid by sorting voxel keys and finds neighbours through a hash table
 * (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:72-119,210-229; extra/cudpp/src/cudpp_hash/hash_multivalue.cpp:45-110;
 * sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:45-114).  The site number of a voxel is its RANK among the occupied voxels in
 * key order (sample, z, y, x) -- and that is a prefix sum of population counts over a bitmap of the batch's bounding box laid
 * out in exactly that order.  One 64-bit word per 32 voxels of an x row: low half = occupancy bits, high half = the number of
 * occupied voxels in front of the word (all scales counted through: subtract the scale's first word's).  Then
 *   site of a voxel       = one 8-byte load + one popcount           (was: hash probes)
 *   the three neighbours (x-1, x, x+1) of a row = the same load      (was: 1-3 probes)
 *   the coarser scale     = OR of 2x2x2 bits, set by the same launch that sets the bits of scale 0 (was: a sort per scale)
 *   site_of_row / order   = rank of every input row, a counting pass and a sort of the (short) per-site segments (was: the sort)
 * Same arrays, bit for bit, as scn_batch_build.  Two read-backs per batch: the bounding box, then the site counts.
 *   scn_occ_bounds   min / max of the coordinates (one launch, one synchronisation); minmax_host = xmin ymin zmin bmin xmax ymax zmax bmax;
 *                    error 3 for coordinates outside the key box, as scn_grid_build
 *   scn_occ_plan     host only: the box (origin aligned to 2^(levels-1) so that scale l is scale 0 shifted), word counts and the
 *                    layout of the buffer; box->bytes = 0 when the box needs more than SCN_OCC_MAX_WORDS words (use scn_batch_build)
 *   scn_occ_chain    bits of every scale + prefix sums into `occ` (box->bytes bytes); n_sites_host (may be NULL: queued only) = the
 *                    site count of every scale after ONE synchronisation (scn_occ_fetch reads them later)
 *   scn_batch_build_occ   as scn_batch_build with a layout from a plan whose flags bit 0 is set; `small` [2 l + 1] bit 1 of scale
 *                    0 set = a voxel holds more than SCN_OCC_SEGMENT_MAX input rows: `order` is NOT sorted inside such sites (the
 *                    caller sorts (site_of_row, row) itself or builds the batch with scn_batch_build).  scratch = NULL: the arena
 *                    holds layout.total bytes, results and scratch; else the arena holds the results (layout.scratch bytes) and
 *                    `scratch` layout.total - layout.scratch bytes that may be freed once the stream has passed the call (the
 *                    neighbour tables of all scales: ~110 bytes per site) */
#define SCN_OCC_MAX_WORDS ((int64_t)1 << 27)
#define SCN_OCC_SEGMENT_MAX 64
typedef struct {
  int32_t levels, nb;                         /* scales; samples (bmax - bmin + 1) */
  int32_t origin[4];                          /* x, y, z, sample of bit 0 at scale 0 */
  int32_t n[SCN_BATCH_MAX_LEVELS][3];         /* voxels per axis (x, y, z) of the box at every scale */
  int32_t wx[SCN_BATCH_MAX_LEVELS];           /* words per x row */
  int64_t woff[SCN_BATCH_MAX_LEVELS + 1];     /* first word of every scale; [levels] = all words */
  int64_t blocks;                             /* 1024-word blocks of the prefix sum */
  int64_t off_info, off_blk, off_words, bytes; /* byte offsets into `occ`; its size */
} scn_occ_box;
int scn_occ_bounds(const void *coords, int coords_dtype, int ncols, int64_t P, int32_t *scratch_dev /* 64 bytes */,
                   int32_t *minmax_host /* [8] */, void *stream);
int scn_occ_plan(const int32_t *minmax, int levels, scn_occ_box *box);
int scn_occ_chain(const void *coords, int coords_dtype, int ncols, int64_t P, const scn_occ_box *box, void *occ, size_t occ_bytes,
                  int32_t *n_sites_host, void *stream);
int scn_occ_fetch(const void *occ, const scn_occ_box *box, int32_t *n_sites_host, void *stream);
int scn_batch_build_occ(const void *coords, const scn_batch_plan *plan_host, const scn_occ_box *box, void *occ, void *arena,
                        size_t arena_bytes, void *scratch, size_t scratch_bytes, void *stream);

/* ------------------------------------------------------- gather-mm-scatter ---
 * out[o,:] = (bias) + sum_k sum_{pairs of list k} in[pair.in_col,:] . Wk
 * Replaces cuda_SubmanifoldConvolution_updateOutput / cuda_Convolution_updateOutput /
 * cuda_Deconvolution_updateOutput (sparseconvnet/SCN/CUDA/Convolution.cpp:36-66,:105-162;
 * sparseconvnet/SCN/CUDA/Deconvolution.cpp:20-49) and the dInput half of the matching *_backward
 * (sparseconvnet/SCN/CUDA/Convolution.cpp:69-102,:165-210).  `out` is fully overwritten.
 *   Ci = planes of `in` (n_in rows; above 4 GiB of input rows a slower fallback kernel runs and says so on stderr), Co = planes of `out` (n_out rows).
 *   Wk[ci][co] = w_transposed ? W[ks][co][ci] : W[ks][ci][co],  ks = w_reverse ? K-1-k : k,
 *   where W is the module's weight [K][Ci][Co] (forward) or [K][Co][Ci] (dInput: w_transposed = 1;
 *   submanifold dInput additionally w_reverse = 1 and the SAME lists, because
 *   (i,o) in list k  <=>  (o,i) in list 26-k).
 *   in_col/out_col: 0/1 for conv + submanifold, 1/0 for deconvolution
 *   (sparseconvnet/SCN/CUDA/Deconvolution.cu:48-49); bases = scn_tile_bases(.., out_col, n_out).
 *   residual (may be NULL): fp32 [n_out, Co] added to the result, out = residual + bias + sum ...: the output tile
 *   starts from its rows, so the shortcut of a residual block (sparseconvnet/networkArchitectures.py:222-229: AddTable over
 *   ConcatTable(shortcut, body)) costs one read instead of a separate elementwise kernel.  May alias nothing else.
 *   colstats (may be NULL): fp32 [scn_conv_stats_blocks(...)][2][Co] -- per block of output rows the column sums and sums
 *   of squares of the RESULT, taken while the output tile is still in LDS / registers; scn_bn_forward_ext finalizes a
 *   batch norm over this tensor from them without reading the tensor once more.  scn_conv_stats_blocks() == 0: this
 *   launch cannot emit them (planes not multiples of 16, > 4 GiB fallback): pass NULL.
 *   counts [K] (device) = pairs per list.  out_unique != 0 promises that EVERY output row appears in exactly one pair of
 *   the rulebook (deconvolution and the dInput of a strided convolution: each fine row has one parity class,
 *   sparseconvnet/SCN/CUDA/Deconvolution.cu:48-49; NetworkInNetwork: the identity list): such launches run without any
 *   accumulation across offsets (k_conv_expand).
 *   wpack: scn_conv_wpack_bytes(K,Ci,Co) bytes holding the weights re-laid out in MFMA fragment order.
 *   wpack_ready = 0: scn_conv_forward packs (W, w_transposed, w_reverse) into it first; 1: the caller already ran
 *   scn_weight_pack for exactly this (W contents, w_transposed, w_reverse) -- weights change once per optimizer
 *   step, not once per call -- and W is not read (planes that are not multiples of 16 always read W itself).
 *   ld_in / ld_out (0 = contiguous): leading dimension in floats of `in` / `out` when they are column slices of wider
 *   row-major tensors -- the two halves of a concatenation (sparseconvnet/tables.py:13-55, JoinTable): a deconvolution then
 *   writes its result straight into the right half of the joined tensor and its backward pass reads the right half of the
 *   joined gradient in place.  ld_in: planes that are multiples of 16 (the pipelined kernels); ld_out: out_unique launches with
 *   planes % 64 == 0 only (scn_conv_plan = 5xx).  residual / colstats / bn_bwd->x stay contiguous [n_out, Co].
 *   tile_passes (may be NULL): scn_tile_passes(bases, K, n_out, ..).  With it -- and with sk_state -- a big launch
 *   (k_conv_tile4) whose tiles would leave 3 % or more of the last round of resident workgroups empty runs as 1024 persistent
 *   workgroups on a static schedule (stream-K): whole tiles first, then ONE slice each of the remaining pass sequence, cut so
 *   that every workgroup ends with the same amount of work.  A tile cut by a slice boundary is continued, not summed: the
 *   workgroup that holds its first passes hands the 128 x 64 partial tile on through the workspace and the next one goes on from
 *   it, so every output element is still accumulated in (offset, chunk, pair) order: results are bit-identical with and
 *   without it.  Needs 4 x 256 workgroups resident at once (checked; otherwise the plain launch).  Faster alone, slower next to
 *   other kernels on the chip: occuseg_amd/SCN.py passes NULL unless SCN.STREAM_K is set.
 *   sk_state (may be NULL): SCN_SK_STATE_BYTES bytes of device memory that hold ZERO when the launch starts; the launch leaves
 *   them zero.  One such block per stream (launches that may overlap need different blocks).  Launches that split their offsets
 *   (the coarse levels) use it too: the last of a tile's workgroups to finish adds the partial tiles itself, in the same fixed
 *   order as the second launch it replaces, and emits the column statistics from the finished tile (scn_conv_stats_blocks
 *   counts on it: pass colstats to such a launch only together with sk_state).
 *   workspace: scn_conv_workspace_bytes(...) bytes.  Launches with few output tiles (the coarse UNet levels) split their
 *   offsets over several workgroups per tile; the partial tiles go to the workspace and are added in a fixed order, so
 *   every launch is bitwise reproducible run to run (the reference scatters with float atomics,
 *   sparseconvnet/SCN/CUDA/Convolution.cu:529,1147).  The big launches (k_conv_tile4) keep the pass table of every
 *   workgroup there (<= 5.3 KB each) and walk it with scalar loads; without a workspace they keep it in LDS. */
size_t scn_conv_wpack_bytes(int K, int Ci, int Co);
size_t scn_conv_workspace_bytes(int K, int64_t stride, int64_t n_in, int64_t n_out, int Ci, int Co, int out_unique);
/* which kernel scn_conv_forward runs for this shape (for per-kernel reporting): 100 * kind + offset split, kind
 * 0 = generic (planes not multiples of 16), 1 = synchronous fallback (> 4 GiB), 2 = k_conv_tile2 (LDS-staged),
 * 4 = k_conv_tile4 (barrier-free), 5 = k_conv_expand (one contribution per output row: deconvolution, conv dInput, NiN) */
int scn_conv_plan(int K, int64_t stride, int64_t n_in, int64_t n_out, int Ci, int Co, int out_unique);
int scn_weight_pack(const float *W, int w_transposed, int w_reverse, int K, int Ci, int Co, void *wpack,
                    void *stream);
/* The re-layout of MANY (weight, orientation) pairs in one launch: what a training loop runs once per optimizer step
 * instead of one scn_weight_pack per layer and direction (the reference reads the module's weight tensor on every
 * call, sparseconvnet/SCN/CUDA/Convolution.cpp:105-162; a caller that caches packed weights must re-pack whenever
 * the weights may have changed -- occuseg_amd/SCN.py does so once per forward pass and after every optimizer step).
 *   descs: DEVICE array of n scn_wpack_desc, first_block = sum of scn_weight_pack_blocks() of the descriptors before it;
 *   total_blocks = that sum over all n.  flags = scn_weight_pack_flags(w_transposed, w_reverse, K, Ci, Co).
 *   Tensors whose planes are not multiples of 16 have 0 blocks (never packed) and must be left out. */
typedef struct scn_wpack_desc {
  const float *W;      /* module weight [K][Ci][Co] (or [K][Co][Ci] with the transposed flag) */
  float *wpack;        /* scn_conv_wpack_bytes(K, Ci, Co) bytes */
  int32_t K, Ci, Co, flags;
  int64_t first_block;
} scn_wpack_desc;
int64_t scn_weight_pack_blocks(int K, int Ci, int Co);
int scn_weight_pack_flags(int w_transposed, int w_reverse, int K, int Ci, int Co);
int scn_weight_pack_batch(const void *descs, int n, int64_t total_blocks, void *stream);
int64_t scn_conv_stats_blocks(int K, int64_t stride, int64_t n_in, int64_t n_out, int Ci, int Co, int out_unique);
/* bn_bwd (may be NULL; needs colstats): this launch is the dInput of a convolution whose INPUT was the output
 * y = leaky(fma(x, w, b)) of a batch norm over x (w = invstd*gamma, b = fma(-mean, w, beta): scn_bn_forward's own
 * arithmetic), i.e. its result d is that batch norm's d_output_features.  colstats then receives, per block of output
 * rows, sum d' and sum (x - mean) d' with d' = d * (y > 0 ? 1 : leakiness) -- the two reductions of
 * BatchNormalization_BackwardPass (sparseconvnet/SCN/CPU/BatchNormalization.cpp:75-87) -- taken while the tile is in LDS /
 * registers; scn_bn_backward_ext finishes the batch norm's backward pass from them without reading x and d once more.
 * x: [n_out, Co] (the rows of the launch's output); weight / bias may be NULL. */
typedef struct scn_bn_bwd_stats {
  const float *x, *save_mean, *save_invstd, *weight, *bias;
  double leakiness;
} scn_bn_bwd_stats;
int scn_conv_forward(const float *in, float *out, const float *W, int w_transposed, int w_reverse,
                     const float *bias, const float *residual, float *colstats, const scn_bn_bwd_stats *bn_bwd,
                     const int32_t *rules,
                     const int32_t *bases, const int32_t *tile_passes, const int32_t *counts,
                     int K, int64_t stride, int in_col, int out_col, int out_unique, int64_t n_in,
                     int64_t n_out, int Ci, int Co, int64_t ld_in, int64_t ld_out, void *wpack, int wpack_ready,
                     void *workspace, size_t workspace_bytes, void *sk_state, void *stream);

/* scn_set_tuning / scn_debug_stamps (kernel-variant selection, timing-only ablation builds, cycle stamps) are NOT
 * part of this ABI: they exist only in libscn_hip_tuning.so (same sources compiled with -DSCN_TUNING), which
 * tools/ and the variant-equivalence test load explicitly. */

/* dW[k] += in[list_k.in_col]^T . dOut[list_k.out_col]   (dW must be zeroed by the caller,
 * as the reference's callers do: sparseconvnet/submanifoldConvolution.py:114-115).
 * ld_in / ld_dout (0 = contiguous): leading dimensions in floats of `in` / `dout` when they are column slices of wider
 * tensors (the right half of a joined gradient); planes must then be multiples of 32.
 * Replaces dConvolution_KMxKN_backward_dW_* (sparseconvnet/SCN/CUDA/Convolution.cu:265-442,:695-753).
 * bases: scn_tile_bases(.., out_col, n_dout) or NULL; with it, narrow inputs (Ci <= 4: the 3 -> m stem) are processed
 * output-row major (every dOut row read once instead of once per pair).
 * workspace (may be NULL): scn_conv_dweight_workspace_bytes(...) bytes.  Without it the workgroups add their partial sums to
 * dW with float atomics (as the reference, Convolution.cu:749): the result is reproducible to rounding only.  With it every
 * workgroup stores its partial sums to a slot of its own and the slots are added to dW in a fixed order: dW is bitwise
 * reproducible run to run.
 * dw_state (may be NULL; used with a workspace only): SCN_DW_STATE_BYTES bytes of device memory that hold ZERO when the launch
 * starts; the launch leaves them zero (arrival tickets).  Launches that may overlap -- different streams -- need different
 * blocks.  With it the fixed-order addition happens INSIDE the dWeight kernel (the last workgroup of every group of 16 slices
 * adds the group, the last group of an (offset, plane block) adds the groups and updates dW): no memset of the workspace, no
 * second launch.  Without it: memset + kernel + reduction kernel.  The two forms add in different (each fixed) orders. */
size_t scn_conv_dweight_workspace_bytes(int K, int64_t stride, int64_t n_in, int64_t n_dout, int Ci, int Co, int has_bases);
int scn_conv_dweight(const float *in, const float *dout, float *dW, const int32_t *rules,
                     const int32_t *counts, const int32_t *bases, int K, int64_t stride, int in_col,
                     int out_col, int64_t n_in, int64_t n_dout, int Ci, int Co, int64_t ld_in, int64_t ld_dout,
                     void *workspace, size_t workspace_bytes, void *dw_state, void *stream);

/* column sums: d_bias[c] = sum_rows dOut[r][c] (sparseconvnet/SCN/CUDA/Convolution.cu:52-75) */
int scn_column_sum(const float *x, float *out, int64_t rows, int C, void *stream);

/* ----------------------------------------------------------- batch norm ---
 * Replaces cuda_BatchNormalization_updateOutput / _backward
 * (sparseconvnet/SCN/CUDA/BatchNormalization.cpp:21-76, kernels sparseconvnet/SCN/CUDA/BatchNormalization.cu:14-199);
 * arithmetic of sparseconvnet/SCN/CPU/BatchNormalization.cpp:12-107.  weight/bias/d_weight/d_bias may be NULL.
 * temp: scn_bn_temp_bytes(C) bytes.  Backward flags: bit 0 (SCN_BN_REWRITE_D_OUT) = rewrite d_out in place with the
 * gradient masked by the (leaky) ReLU, as the reference does (BatchNormalization.cu:151-153); without it d_out is left
 * untouched and the masked gradient is staged in d_x instead (same traffic, and an autograd caller needs no private
 * copy).  Bit 1 (SCN_BN_ACCUMULATE_DPARAMS) = d_weight / d_bias += instead of = (the reference assigns).
 * Bit 2 (SCN_BN_MASK_FROM_X) = do not read y at all (it may be NULL): the sign of y is recomputed from x with the forward
 * pass's own arithmetic (y = fma(x, w, b), w = invstd*gamma, b = fma(-mean, w, beta): bit-identical), d_out is read
 * twice and never written: 20 B per element instead of 28.  Needs `bias`; excludes bit 0. */
#define SCN_BN_REWRITE_D_OUT 1
#define SCN_BN_ACCUMULATE_DPARAMS 2
#define SCN_BN_MASK_FROM_X 4
size_t scn_bn_temp_bytes(int C);
int scn_bn_forward(const float *x, float *y, float *save_mean, float *save_invstd,
                   float *running_mean, float *running_var, const float *weight,
                   const float *bias, double eps, double momentum, int train, double leakiness,
                   int64_t N, int C, void *temp, void *stream);
/* training forward with the statistics handed over by the producing kernels: nseg segments of planes (in order, covering
 * all C), segment i = seg_ncols[i] planes with partial sums seg_partial[i] = [seg_nblk[i]][2][seg_ncols[i]] (device;
 * the three arrays themselves are HOST arrays).  Only finalize + apply run: 8 instead of 12 bytes per element.
 * state (may be NULL; here and in scn_bn_backward_ext): the stream's control block (SCN_SK_STATE_BYTES of device memory that hold
 * ZERO between launches, shared with scn_conv_forward's sk_state: launches of one stream never overlap).  With it, more than 256
 * partials are folded AND finalized by one launch (tickets in words 32 .. 32 + ceil(C / 128) - 1, one per group of 128 planes,
 * zero again afterwards: the SAME words scn_conv_forward's split launches use for their tile tickets and stream-K for its slab
 * flags) instead of two; same arithmetic in the same order.  One stream per control block: two launches that name one block
 * must never run at the same time, and a launch that dies after drawing tickets leaves the block unusable (zero it again). */
int scn_bn_forward_ext(const float *x, float *y, float *save_mean, float *save_invstd,
                       float *running_mean, float *running_var, const float *weight,
                       const float *bias, double eps, double momentum, double leakiness,
                       int64_t N, int C, void *temp, int nseg, const int32_t *seg_ncols_host,
                       const float *const *seg_partial_host, const int32_t *seg_nblk_host,
                       void *state, void *stream);
int scn_bn_backward(const float *x, float *d_x, const float *y, float *d_out,
                    const float *save_mean, const float *save_invstd, const float *weight,
                    const float *bias, float *d_weight, float *d_bias, double leakiness, int64_t N,
                    int C, int flags, void *temp, void *stream);
/* The SCN_BN_MASK_FROM_X backward pass with two things handed in (both may be NULL / 0):
 *   partial [nblk][2][C]: the two reductions, emitted by the kernel that produced d_out (scn_conv_forward's bn_bwd):
 *   the statistics pass is skipped, d_out and x are read once (12 instead of 20 bytes per element);
 *   d_x_add [N, C] (leading dimension ld_add floats, 0 = C: it may be the left half of a joined gradient): added to the result, d_x = (batch norm's d_input) + d_x_add -- the gradient arriving at x along
 *   another path (the shortcut of a residual block), which autograd would otherwise add with a kernel of its own.
 * flags: SCN_BN_ACCUMULATE_DPARAMS only. */
int scn_bn_backward_ext(const float *x, float *d_x, const float *d_out, const float *save_mean,
                        const float *save_invstd, const float *weight, const float *bias, float *d_weight,
                        float *d_bias, double leakiness, int64_t N, int C, int flags, void *temp,
                        const float *partial, int nblk, const float *d_x_add, int64_t ld_add, void *state, void *stream);

/* ------------------------------------------------------ input / output ---
 * Replaces InputLayer_fp/bp (sparseconvnet/SCN/CUDA/IOLayers.cu:17-76) and the four wrappers of
 * sparseconvnet/SCN/CUDA/IOLayers.cpp:17-154.  average != 0 for mode 4. */
int scn_input_forward(const float *feats, float *out, const int32_t *order,
                      const int32_t *site_start, int64_t N, int C, int average, void *stream);
int scn_input_backward(float *d_feats, const float *d_out, const int32_t *site_of_row,
                       const int32_t *site_start, int64_t P, int C, int average, void *stream);
int scn_output_forward(const float *in, float *out, const int32_t *site_of_row, int64_t P,
                       int C, void *stream);
int scn_output_backward(float *d_in, const float *d_out, const int32_t *order,
                        const int32_t *site_start, int64_t N, int C, void *stream);

/* ------------------------------------------ either side of the hot path ---
 * Normal-guided filters (the fork's use_normal path).  site_normals [N,3] = normalised mean of the point normals of
 * each site, site_ori [N] = OrientedFilter(normal) in {0, 2, 4}.  Replaces the host loop of
 * cuda_InputLayer_updateOutput (sparseconvnet/SCN/CUDA/IOLayers.cpp:36-65), Float3::normalize
 * (sparseconvnet/SCN/Metadata/Metadata.h:95-101) and OrientedFilter (sparseconvnet/SCN/Metadata/RectangularRegions.h:12-32). */
int scn_site_normals(const float *point_normals, const int32_t *order, const int32_t *site_start,
                     int64_t N, float *site_normals, int32_t *site_ori, void *stream);

/* out[q] = site id (key rank) of voxel query[q] / div in the grid with sorted unique keys site_keys, or -1.
 * query int32 [Q,ncols] (ncols 4: last column = sample; ncols 3: all in sample `sample`).  Replaces
 * ResolutionBasedScatteringCuda (sparseconvnet/SCN/Metadata/ConvolutionRules.h:327-342, bound at
 * sparseconvnet/SCN/pybind.cpp:33-36). */
int scn_lookup_sites(const uint64_t *site_keys, int64_t N, const int32_t *query, int ncols, int64_t Q,
                     int div, int sample, int32_t *out, void *stream);

/* Segment reductions over instance labels (the loss side, examples/ScanNet/train_instance.py:203-242 and
 * examples/ScanNet/discriminative.py:52-53,146 call torch_scatter 1.4.0's scatter_add / scatter_mean / scatter_std /
 * scatter_max with dim = 0): src fp32 [P,C], seg int64 [P] with labels in [0,S), out fp32 [S,C].
 *   scn_segment_reduce: out[seg[p],c] (op 0: +=, op 1: = max) src[p,c]; `out` must hold 0 (op 0) or -FLT_MAX (op 1);
 *   scn_segment_count:  counts[s] += |{p : seg[p] = s}|   (fp32, `counts` zeroed by the caller);
 *   scn_segment_gather: out[p,c] = table[seg[p],c] * (scale ? scale[seg[p]] : 1);
 *   scn_segment_sqdev:  out[seg[p],c] += (src[p,c] - mean[seg[p],c])^2. */
int scn_segment_reduce(const float *src, const int64_t *seg, int64_t P, int C, int S, int op, float *out,
                       void *stream);
int scn_segment_count(const int64_t *seg, int64_t P, int S, float *counts, void *stream);
int scn_segment_gather(const float *table, const int64_t *seg, const float *scale, int64_t P, int C,
                       int S, float *out, void *stream);
int scn_segment_sqdev(const float *src, const float *mean, const int64_t *seg, int64_t P, int C, int S,
                      float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SCN_HIP_H */
