"""Drop-in for the reference's pybind11 module ``sparseconvnet.SCN``
(sparseconvnet/SCN/pybind.cpp:11-239, declarations sparseconvnet/SCN/sparseconvnet.h), D = 3,
hot-path functions only.  Same names, argument order and ownership rules as the
reference: callers pass empty outputs which are ``resize_``d and filled here;
``d_weight`` / ``d_bias`` arrive zeroed and are accumulated into; batch-norm
backward rewrites ``d_output_features`` in place.

Everything here is host-side plumbing (torch owns device memory and streams);
all arithmetic and all rulebook construction happens in libscn_hip.so through
the C ABI of include/scn_hip.h.  CPU tensors are refused: there is no CPU path.
"""
import contextlib
import os
import ctypes
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream

GRANULE = 32  # SCN_TILE_GRANULE

# When set to a list, every gather-mm launch appends (kernel, flops, bytes, start_event, end_event):
# used by bench.py to time the dominant kernels with HIP events on the launch stream.
PROFILE = None
PROFILE_KINDS = ("mm", "bn", "rules", "io")  # which launches carry events: matrix kernels / batch norm / rule build / io
# kernel name -> bytes the profiled launches read ON TOP of SURVEY 8d's formula because something is fused into them: the
# shortcut's rows (`residual`) and, for a dInput launch that takes a batch norm's backward sums, that batch norm's x.  bench.py
# prints them beside the algorithmic bytes: traffic above (algorithmic + fused) is re-reading, traffic above algorithmic alone is not.
PROFILE_FUSED = {}


_EVENT_POOL = []  # timing events created ahead of the timed region (creating one inside it costs far more than recording)


def warm_event_pool(n):
    """create (and record once, which is what really creates them) n timing events for _prof_begin/_prof_end to use"""
    while len(_EVENT_POOL) < n:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _EVENT_POOL.append(ev)


def _event():
    return _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)


def _prof_begin(kind="mm"):
    if PROFILE is None or kind not in PROFILE_KINDS:
        return None
    ev = _event()
    ev.record()
    return ev


def _prof_end(ev, kernel, flops, nbytes, shape=None):
    """nbytes may be a callable (evaluated when the profile is summarised: pair counts that are not known on the
    host yet when a rulebook build is queued)"""
    if ev is not None:
        end = _event()
        end.record()
        PROFILE.append((kernel, flops, nbytes, ev, end, shape))


_CONV_KERNELS = {0: "conv_generic", 1: "conv_tile_fallback", 2: "conv_tile2", 4: "conv_tile4", 5: "conv_expand"}


def _lib_():
    return _lib.load()


_SIZE_KEYS = {}  # id(tensor) -> (tensor, version, key): the modules pass the same few size tensors on every call


def _size_key(t):
    if not torch.is_tensor(t):
        return tuple(int(v) for v in t)
    e = _SIZE_KEYS.get(id(t))
    if e is not None and e[0] is t and e[1] == t._version:
        return e[2]
    key = tuple(int(v) for v in t.tolist())
    if len(_SIZE_KEYS) > 4096:
        _SIZE_KEYS.clear()
    _SIZE_KEYS[id(t)] = (t, t._version, key)
    return key


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "occuseg_amd.SCN.%s: features must live on the GPU -- this backend has no CPU path "
            "(the CPU restatement lives in oracle/ and is test infrastructure only)" % what)
    if t.dtype != torch.float32:
        raise RuntimeError("occuseg_amd.SCN.%s: only float32 features are supported" % what)


# Rulebook requests seen during the previous forward pass, keyed by the input spatial size.  The
# reference builds every rulebook lazily at first use (sparseconvnet/SCN/Metadata/Metadata.cpp:504-529,
# :598-625); doing that here would put a stream synchronisation (site counts are data dependent) in
# the middle of the queued convolutions of every level.  Instead the next InputLayer call replays
# the recorded requests up front, while the queue is still empty, so the host runs ahead of the GPU
# for the rest of the step.  Purely a scheduling matter: the lazily built result is identical.
_PLANS = {}
_PLANS_USED = {}  # spatial size -> the request list of the most recent forward pass that really ran (tools, bench)
PREBUILD = True
# stream-K for the big gather-mm launches (k_conv_tile4): every rulebook carries the pass prefix of its 128-row tiles and the
# slice tables (scn_tile_passes); handed those and a control block, a launch whose tiles would leave 3 % or more of the last
# round of resident workgroups empty runs as 1,024 persistent workgroups on a static schedule: whole tiles first, then one slice of
# the remaining pass sequence each, cut to fit -- bit-identical results.  Alone such a launch is 1-13 % faster
# (tools/ab_stream_k.py).  Inside the train step that holds for the launches of the FORWARD direction, which have the chip to
# themselves: 43.25 -> 42.81 ms per step (profiles/r04_ab_stream_k_fwd.txt).  In the backward pass persistent workgroups hold
# every CU's LDS for the whole launch, so the dWeight kernels of the side stream and the next batch's rule build, which plain
# launches interleave with, queue behind them or delay the workgroups they displace, and every slice waits for the slowest
# (45.2 ms with stream-K everywhere; DESIGN.md section 3.3e, profiles/r04_ab_stream_k.txt).
# SCN_STREAM_K=fwd (default): forward launches only; =1: every launch; =0: off.  SCN.STREAM_K / SCN.STREAM_K_FWD_ONLY at run time.
# One host read-back per batch for the site counts of ALL scales (scn_grid_chain_counts) instead of one per scale; the
# scales are those the previous forward pass on the same input size asked for (PREBUILD's plan).  Measured (m64 batch alone,
# tools/rules_alone.py, profiles/r04_rules_timeline.txt): the six read-backs cost 0.31 ms together (span 2.45 ms, kernels
# 2.14 ms) and the counting launch that replaces five of them 0.50 ms (1.5 M compare-and-swaps into hash sets): 2.60
# against 2.45 ms per batch -- the build is bound by its kernels and its ~185 launches, not by the read-backs.  But with ONE
# early read-back the host issues the rest of the build and the forward pass behind it without waiting: the forward-only
# configuration, which builds its rulebooks in the forward pass and is bound by host time, gains 4-8 % (cfg2 4.56 -> 4.18,
# 4.23 -> 4.06 ms).  "auto" (default): the chain for builds inside a forward pass or under torch.no_grad(), per-scale read-backs
# for builds that prefetch_input runs on its side stream beside a backward pass (there the host has nothing else to issue and
# the device time counts).
# One C call per batch for every grid and rulebook (scn_batch_build, round 5) where the site counts of all scales are known up front
# (the grid chain): the forward-only configuration spent a third of its host time in ~60 ctypes calls and ~120 allocations
BATCH_BUILD = os.environ.get("SCN_BATCH_BUILD", "1") != "0"
# The batch build from occupancy bitmaps (scn_batch_build_occ: no sort, no hash map; include/scn_hip.h).  "auto": when the batch's
# bounding box is no larger than OCC_WORDS_PER_ROW 64-bit words per input row (a compact cloud; a handful of far-apart points would
# pay for memsets and prefix sums over an empty box), True: whenever the box fits SCN_OCC_MAX_WORDS, False: never.
_OB = os.environ.get("SCN_OCC_BUILD", "auto")
OCC_BUILD = True if _OB == "1" else False if _OB == "0" else "auto"
OCC_WORDS_PER_ROW = 64
OCC_FIRST_BATCH = os.environ.get("SCN_OCC_FIRST_BATCH", "1") != "0"  # see Metadata_3.inputLayer ("speculative")
OCC_FIRST_BATCH_LEVELS = 8
OCC_STATS = {"built": 0, "too_large": 0, "order_repaired": 0}  # since import (tests, tools)
DEFER_PREFETCH_READBACK = os.environ.get("SCN_DEFER_PREFETCH_READBACK", "1") != "0"
_PINNED = []  # pinned host buffers of the deferred read-backs (allocating one costs a millisecond: kept and re-used)


def _pinned(n, dtype):
    for i, t in enumerate(_PINNED):
        if t.dtype == dtype and t.numel() >= n:
            return _PINNED.pop(i)
    return torch.empty(max(n, 1024), dtype=dtype, pin_memory=True)

_GC = os.environ.get("SCN_GRID_CHAIN", "auto")
GRID_CHAIN = True if _GC == "1" else False if _GC == "0" else "auto"
_GRID_CHAIN_MAX = 12
STREAM_K = os.environ.get("SCN_STREAM_K", "fwd") in ("1", "fwd")
STREAM_K_FWD_ONLY = os.environ.get("SCN_STREAM_K", "fwd") == "fwd"
# launches that split their offsets over several workgroups per tile (the coarse levels): the last workgroup of a tile adds the
# partial tiles itself (scn_conv_forward's sk_state) instead of a second launch, and emits the batch norm's statistics.
# False: k_split_reduce as in round 3 (A/B: tools/ab_step.py SPLIT_FUSED=0,1)
SPLIT_FUSED = os.environ.get("SCN_SPLIT_FUSED", "1") != "0"
# the statistics a convolution hands to the batch norm next to it (one partial per 128-row tile: thousands) are folded AND finalized by
# one launch -- a ticket in the stream's control block -- instead of two (A/B: tools/ab_step.py BN_FOLD_FUSED=0,1)
BN_FOLD_FUSED = os.environ.get("SCN_BN_FOLD_FUSED", "1") != "0"
_SK_STATE = {}  # (device index, stream handle) -> the zeroed control block of that stream's stream-K launches


def _sk_state(device, stream_handle=None):
    """SCN_SK_STATE_BYTES of zeros per (device, stream): slice counters and slab flags of a stream-K launch.  Every launch
    leaves the block as it found it; launches on ONE stream never overlap, so they can share one."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream if stream_handle is None else stream_handle)
    t = _SK_STATE.get(key)
    if t is None:
        t = _SK_STATE[key] = torch.zeros(_lib.SK_STATE_BYTES, dtype=torch.uint8, device=device)
    return t


_SK_WANTED = set()   # (lists of the rulebook, output column) of the stream-K launches seen so far
_SK_MIN_TILES = 512  # (k_conv_tile4 takes launches of >= 512 tiles)


def _prefetch_in_flight():
    """A rulebook prefetch of this process that no InputLayer call has adopted yet and whose kernels may still be running (a
    serving loop issues the next scene's prefetch BEFORE the current forward pass).  The static schedule of a stream-K launch
    needs its 1,024 workgroups resident at once; scn's residency check (sk_all_resident) asks for an EMPTY chip, and with part
    of the chip held by the prefetch stream's kernels the launch would lean on in-order dispatch to make progress (round-4
    advisor): such a launch runs one workgroup per tile instead.  (A training step's prefetch is issued behind the backward
    pass and adopted -- the main stream waits for its event -- before the next forward launch: never in flight here.)"""
    for ent in _PREFETCHED.values():
        if not ent[3].query():
            return True
    return False


def _tile_passes(rb, out_col, n_out):
    lib = _lib_()
    passes = torch.empty(lib.scn_tile_passes_bytes(n_out) // 4, dtype=torch.int32, device=rb.rules.device)
    check(lib.scn_tile_passes(ptr(rb.bases[out_col]), rb.K, n_out, ptr(passes), stream()))
    rb.passes[out_col] = passes
    return passes


def stream_k_gave_up():
    """True if a stream-K launch of this process ever stopped waiting for another workgroup's partial tile (word 9 of the
    control block; a bounded wait, so that a broken launch cannot hang the GPU): its result is then wrong.  Never observed.
    The default path does not depend on anybody asking: every batch's read-back carries the words (_sk_health_words) and
    Metadata_3.inputLayer raises."""
    return any(bool(t.view(torch.int32)[9].item()) for t in _SK_STATE.values())


def _sk_health_words(device):
    """word 9 of every control block on `device`, as one-element int32 tensors for the batch's read-back (torch.cat)"""
    return [t.view(torch.int32)[9:10] for (idx, _), t in _SK_STATE.items() if idx == device.index]


def _sk_check(words, device):
    """`words`: the values of _sk_health_words at the batch's read-back.  A stream-K piece that gave up computed from a slab that
    had not been published: that launch's output is wrong, and the control block is poisoned -- the late producer raises the
    flag its reader has already reset, so every later launch on that stream (the tickets of the split launches share the block)
    would read stale slabs without waiting.  Hard error (round-4 advisor): the blocks are zeroed behind a device-wide
    synchronisation and the caller is told."""
    if not any(int(w) for w in words):
        return
    torch.cuda.synchronize(device)
    for (idx, _), t in _SK_STATE.items():
        if idx == device.index:
            t.zero_()
    torch.cuda.synchronize(device)
    raise RuntimeError(
        "occuseg_amd: a stream-K launch of k_conv_tile4 stopped waiting for another workgroup's partial tile (sk_state word 9): "
        "the output of that launch -- hence of the step it belonged to -- is wrong.  The control blocks have been reset; discard "
        "the step.  SCN_STREAM_K=0 turns stream-K off (the persistent workgroups were not all resident: another process or a "
        "lower-priority launch held part of the chip for over a second)")


# The tickets of the split launches, of the batch norm's fold and of the deterministic dWeight share their control blocks with
# stream-K's flags, and every launch must leave its block ZERO.  Word 9 (above) only tells of a stream-K wait that gave up; a
# launch that died after drawing tickets leaves other words behind, and every later launch on that stream would then sum
# incomplete slabs or never finalize -- silently (round-5 advisor).  check_control_blocks is the cheap sentinel: at a point
# where the CURRENT stream has no launch in flight on its own block (the start of a train step: FlatGradBucket.zero()), "is any
# byte of this stream's blocks non-zero" is computed on the device, copied to pinned memory without waiting, and LOOKED AT by
# the next call -- one step late, no host synchronisation, two tiny launches per step.
_BLOCK_CHECKS = {}  # device index -> (pinned flag tensor, event) of the check queued by the previous call


def check_control_blocks(device=None):
    device = _norm_device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    prev = _BLOCK_CHECKS.pop(device.index, None)
    if prev is not None:
        host, ev = prev
        ev.synchronize()  # (queued a whole step ago: long done)
        bad = bool(host[0].item())
        _PINNED.append(host)
        if bad:
            torch.cuda.synchronize(device)
            for (idx, _), t in list(_SK_STATE.items()) + list(_DW_TICKETS.items()):
                if idx == device.index:
                    t.zero_()
            torch.cuda.synchronize(device)
            raise RuntimeError(
                "occuseg_amd: a control block (tickets of the split / batch-norm / dWeight launches, stream-K flags) did not return "
                "to zero: a launch on that stream died or was aborted half way, and every later launch that draws tickets from the "
                "block would have summed incomplete partial results.  The blocks have been reset; discard the last step.")
    # (every block of the device: the caller has joined the dWeight side stream -- FlatGradBucket.zero() calls dw_join() first --
    #  and no other stream of this process runs launches that draw tickets)
    mine = [t for (idx, _), t in list(_SK_STATE.items()) + list(_DW_TICKETS.items()) if idx == device.index]
    if not mine:
        return
    flag = torch.stack([t.any() for t in mine]).any().to(torch.uint8).reshape(1)
    host = _pinned(1, torch.uint8)
    host[:1].copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _BLOCK_CHECKS[device.index] = (host, ev)


KEEP_NBR = False  # tests only: keep each submanifold rulebook's neighbour table (`d_global_query_result`) alive
_COORD_DTYPES = {torch.int32: 0, torch.int64: 1, torch.float32: 2}  # scn_grid_build coords_dtype

# Grids + rulebooks built AHEAD of the InputLayer call that will use them (prefetch_input below), keyed by the
# identity of the coordinate tensor.  Pipelining across steps only: what is built is what inputLayer would build.
_PREFETCHED = {}
_SIDE_STREAMS = {}


class _Grid(object):
    """Active sites of one scale (device resident)."""
    __slots__ = ("n", "coords", "keys", "hash_keys", "hash_vals", "hash_cap")

    def __init__(self, n, coords, keys):
        self.n = n
        self.coords = coords  # int32 [N,4]  (x,y,z,b) in global site order
        self.keys = keys      # int64 [N]    (b<<31 | key31), ascending
        self.hash_keys = None
        self.hash_vals = None
        self.hash_cap = 0


class _RuleBook(object):
    """Device rulebook: rules int32 [K, stride, 2], counts int32 [K] (device),
    tile-base tables per output column."""
    __slots__ = ("K", "stride", "rules", "counts", "bases", "passes", "n_rows", "total_pairs", "nbr", "unique_col", "ori",
                 "class_books")

    def __init__(self, K, stride, rules, counts):
        self.K = K
        self.stride = stride
        self.rules = rules
        self.counts = counts
        self.bases = {}      # out_col -> int32 [K, n_gran+1]
        self.passes = {}     # out_col -> int32 [tiles of 128 rows + 1]: pass prefix of the tiles for stream-K launches (may be absent)
        self.n_rows = {}     # out_col -> number of output rows
        self.total_pairs = None
        self.nbr = None
        self.unique_col = None  # column in which every row id 0..n-1 appears exactly once over all lists (2: both)
        self.ori = None          # normal-guided rulebook: int32 [N] orientation class of every (output) site
        self.class_books = None  # ... and, built when a dInput needs them, {class: plain rulebook of that class's outputs}

    def host_lists(self):
        """reference-format host rulebook: list of K int32 arrays [R_k, 2].  A normal-guided rulebook is kept sorted by
        output site on the device; the reference's list n is the same pairs as up to three runs, in the order of the
        ORIGINAL offset i they were found at (remap_rules_with_normal walks the old lists i = 0..26 and appends,
        sparseconvnet/SCN/Metadata/SubmanifoldConvolutionRules.h:232-243): reproduced here by a stable sort on i."""
        counts = self.counts.cpu().tolist()
        r = self.rules.cpu().numpy()
        lists = [r[k, :counts[k]].copy() for k in range(self.K)]
        if self.ori is not None:
            ori = self.ori.cpu().numpy()
            inv = np.empty((6, 27), np.int64)
            for o in range(6):
                inv[o, ROT_INDEX[o]] = np.arange(27)
            for n in range(self.K):
                old_i = inv[ori[lists[n][:, 1]], n]
                lists[n] = lists[n][np.argsort(old_i, kind="stable")]
        return lists


# rot_index / the `index` table of remap_rules_with_normal (sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:8-15,
# sparseconvnet/SCN/Metadata/SubmanifoldConvolutionRules.h:217-223): new list = ROT_INDEX[orientation class][old offset]
ROT_INDEX = np.array([
    0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,
    24,25,26,21,22,23,18,19,20,15,16,17,12,13,14,9,10,11,6,7,8,3,4,5,0,1,2,
    6,7,8,15,16,17,24,25,26,3,4,5,12,13,14,21,22,23,0,1,2,9,10,11,18,19,20,
    18,19,20,9,10,11,0,1,2,21,22,23,12,13,14,3,4,5,24,25,26,15,16,17,6,7,8,
    2,11,20,5,14,23,8,17,26,1,10,19,4,13,22,7,16,25,0,9,18,3,12,21,6,15,24,
    18,9,0,21,12,3,24,15,6,19,10,1,22,13,4,25,16,7,20,11,2,23,14,5,26,17,8], np.int64).reshape(6, 27)


class Metadata_3(object):
    """Mirror of ``Metadata<3>`` (sparseconvnet/SCN/Metadata/Metadata.h:218-364): per-scale grids,
    rulebook caches keyed like the reference (sparseconvnet/SCN/Metadata/32bits.h:17-76)."""

    def __init__(self):
        self.normal_guide_scale = 10240
        self.clear()

    def clear(self):
        self.grids = {}
        self.nActive = {}
        self.normals = {}   # spatial size -> (site normals fp32 [N,3], orientation class int32 [N])  (use_normal path)
        self.submanifoldRuleBooks = {}
        self.ruleBooks = {}
        self.device = None
        self.inputSpatialSize = None
        self.input_mode = None
        self.input_P = 0
        self.input_order = None
        self.input_site_of_row = None
        self.input_site_start = None
        self._plan = []
        self._deferred = None     # (pinned host copy of a prefetched build's read-back, its event, what to do with it)
        self._chain_counts = {}   # spatial size -> N from the grid chain (scn_grid_chain_counts), this batch
        self._occ = None          # (scn_occ_box, buffer) between scn_occ_chain and scn_batch_build_occ
        self._grid_status = []    # (device [N, error flag] of a grid built without read-back, the N it was built for)

    def setNormalGuideScale(self, s):
        self.normal_guide_scale = int(s)

    # -- grids ---------------------------------------------------------------
    def _build_grid(self, coords_dev, dtype_code, ncols, shift, P, n_known=None):
        lib = _lib_()
        dev = coords_dev.device
        ev = _prof_begin("rules")
        tb = lib.scn_grid_build_temp_bytes(P)
        temp = torch.empty(tb, dtype=torch.uint8, device=dev)
        keys_sorted = torch.empty(P, dtype=torch.int64, device=dev)
        order = torch.empty(P, dtype=torch.int32, device=dev)
        site_of_row = torch.empty(P, dtype=torch.int32, device=dev)
        site_start = torch.empty(P + 1, dtype=torch.int32, device=dev)
        if n_known is None:
            n = ctypes.c_int32(0)
            check(lib.scn_grid_build(ptr(coords_dev), int(dtype_code), ncols, shift, P, ptr(temp), tb,
                                     ptr(keys_sorted), ptr(order), ptr(site_of_row), ptr(site_start),
                                     ctypes.byref(n), stream()))
            N = int(n.value)
        else:  # counted by the grid chain: no read-back here, the build's own count is compared at the batch's next one
            status = torch.empty(2, dtype=torch.int32, device=dev)
            check(lib.scn_grid_build_async(ptr(coords_dev), int(dtype_code), ncols, shift, P, ptr(temp), tb,
                                           ptr(keys_sorted), ptr(order), ptr(site_of_row), ptr(site_start),
                                           ptr(status), stream()))
            # (a callable: the chain's counts are fetched only now, with this grid's launches -- sized by P alone -- queued)
            N = int(n_known() if callable(n_known) else n_known)
            self._grid_status.append((status, N))
        site_coords = torch.empty((N, 4), dtype=torch.int32, device=dev)
        site_keys = torch.empty(N, dtype=torch.int64, device=dev)
        check(lib.scn_grid_finalize(ptr(coords_dev), int(dtype_code), ncols, shift, P, N, ptr(keys_sorted),
                                    ptr(order), ptr(site_start), ptr(site_coords), ptr(site_keys),
                                    stream()))
        # algorithmic bytes: coordinates read twice (keys, site coords), keys + order written and read by the ranking
        # passes, site_of_row / site_start / site coords / site keys written (the radix sort's own passes not counted)
        _prof_end(ev, "grid_build", 0.0, 2.0 * P * ncols * coords_dev.element_size() + P * (12 + 12 + 8) + N * 24.0,
                  (P, N))
        return _Grid(N, site_coords, site_keys), order, site_of_row, site_start[:N + 1]

    def _hash(self, g):
        if g.hash_keys is None:
            lib = _lib_()
            cap = lib.scn_hash_capacity(g.n)
            g.hash_keys = torch.empty(cap, dtype=torch.int64, device=g.coords.device)
            g.hash_vals = torch.empty(cap, dtype=torch.int32, device=g.coords.device)
            g.hash_cap = cap
            check(lib.scn_hash_build(ptr(g.keys), g.n, ptr(g.hash_keys), ptr(g.hash_vals), cap, stream()))
        return g

    def inputLayer(self, spatialSize, coords, batchSize, mode, device, normals=None, _in_prefetch=False):
        """Metadata::inputLayer (sparseconvnet/SCN/Metadata/Metadata.cpp:424-446) ->
        inputLayerRulesSimple (sparseconvnet/SCN/Metadata/IOLayersRules.h:135-202).  ``normals`` (float [P,3], one per
        input point) switches on the fork's normal-guided filters: per-site normals are averaged here
        (sparseconvnet/SCN/CUDA/IOLayers.cpp:36-65) BEFORE any rulebook of this batch is built."""
        if mode not in (3, 4):
            raise NotImplementedError("InputLayer: the reference GPU path supports mode 3 and 4 only "
                                      "(sparseconvnet/SCN/Metadata/IOLayersRules.h:141)")
        assert coords.dim() == 2 and coords.size(1) in (3, 4)
        # int64 as the reference passes them, or int32 / float32 straight from a collate function (converted on
        # the GPU like the reference's `.type(torch.LongTensor)`, sparseconvnet/ioLayers.py:56)
        assert coords.dtype in _COORD_DTYPES, coords.dtype
        key = _size_key(spatialSize)
        _WEPOCH[0] += 1  # a new batch: packed weights are re-made from the module's weights at their next use
        in_prefetch = bool(_in_prefetch)
        if not in_prefetch:
            _step_begins(_norm_device(device))  # (the coarse-level mark of the step before is stale)
        pre = _PREFETCHED.pop(id(coords), None)
        if pre is not None and pre[0] is coords and pre[4] is normals \
                and pre[1] == (key, int(mode), _norm_device(device), self.normal_guide_scale):
            self._adopt(pre[2], pre[3])
            return
        self.clear()
        self.device = device
        self.inputSpatialSize = key
        c = coords.to(device, non_blocking=True).contiguous()
        P = c.size(0)
        old_plan = _PLANS.get(key, [])
        # (normals: the batch build leaves the input scale's 3x3x3 rulebook -- the one they re-sort, remap_rules_with_normal -- to
        #  getSubmanifoldRuleBook below; where they would guide STRIDED rules too, nothing is built ahead and the op says what the
        #  reference's path for that is worth, see getRuleBook)
        batchable = (BATCH_BUILD and (normals is None or key[0] < self.normal_guide_scale) and not KEEP_NBR
                     and not (PROFILE is not None and "rules" in PROFILE_KINDS))
        speculative = False
        if not old_plan and PREBUILD and OCC_BUILD and OCC_FIRST_BATCH and batchable:
            # the FIRST batch on this input size: no forward pass has said yet which rulebooks it wants.  From bitmaps every grid and
            # rulebook a UNet could ask for (3x3x3 at every scale, 2/2 between scales, while the size halves) costs a fraction of
            # building the ones it does ask for by sort + hash, one at a time; what the ops then do not ask for is dropped with
            # this batch (the plan of the next one is what they DID ask for).  A box too large for that falls through to the old way.
            # (OCC_FIRST_BATCH_LEVELS scales: the box's origin is aligned to 2^(scales - 1) -- with all twelve a cloud far from
            #  the origin would get a box that starts at 0; a network deeper than that builds its coarsest grids as it asks)
            old_plan, sz = [], key
            while len(old_plan) < 2 * OCC_FIRST_BATCH_LEVELS - 1:
                old_plan.append(("sub", sz, (3, 3, 3)))
                if any(v % 2 or v < 4 for v in sz) or len(old_plan) >= 2 * OCC_FIRST_BATCH_LEVELS - 1:
                    break
                nxt = tuple(v // 2 for v in sz)
                old_plan.append(("conv", sz, nxt, (2, 2, 2), (2, 2, 2)))
                sz = nxt
            speculative = True
        # auto: inside a forward pass, and in a prefetch issued under torch.no_grad() (a serving loop: the same host thread
        # issues the forward pass, nothing hides a read-back); not in a prefetch beside a backward pass
        chain = GRID_CHAIN if GRID_CHAIN != "auto" else (not in_prefetch or not torch.is_grad_enabled())
        fetch, sizes = self._chain(c, key, old_plan if PREBUILD and (chain or (OCC_BUILD and batchable)) else [],
                                   occ=OCC_BUILD if batchable else False, hashed=chain and not speculative)
        if speculative and fetch is None:
            old_plan = []  # (nothing was built ahead: the ops build what they ask for)
        self.input_mode = int(mode)
        self.input_P = P
        batch_small = None
        if fetch is not None and batchable:
            # every grid and rulebook of the chain queued by ONE call into one arena (scn_batch_build)
            batch_small = self._batch_build(c, sizes, fetch(), old_plan, skip_sub=(key,) if normals is not None else ())
            g = self.grids[key]
        else:
            g, order, site_of_row, site_start = self._build_grid(c, _COORD_DTYPES[c.dtype], c.size(1), 0, P,
                                                                 n_known=(lambda: fetch()[key]) if fetch is not None else None)
            self.grids[key] = g
            self.nActive[key] = g.n
            self.input_order, self.input_site_of_row, self.input_site_start = order, site_of_row, site_start
        if normals is not None:
            assert normals.dim() == 2 and normals.size(0) == P and normals.size(1) == 3, normals.shape
            pn = normals.to(device=device, dtype=torch.float32).contiguous()
            sn = torch.empty((g.n, 3), dtype=torch.float32, device=c.device)
            ori = torch.empty(g.n, dtype=torch.int32, device=c.device)
            check(_lib_().scn_site_normals(ptr(pn), ptr(self.input_order), ptr(self.input_site_start), g.n, ptr(sn), ptr(ori), stream()))
            self.normals[key] = (sn, ori)
        self._plan = []          # requests made by the ops of THIS forward pass (stale ones drop out)
        _PLANS[key] = self._plan
        if PREBUILD and old_plan:
            pending = []
            from_batch = {id(rb) for _, rb in batch_small[1][2]} if batch_small is not None else ()
            for req in old_plan:
                try:
                    if req[0] == "sub":
                        rb = self.getSubmanifoldRuleBook(req[1], req[2], _defer=True)
                        if rb.total_pairs is None and id(rb) not in from_batch:  # (a rulebook of the batch build: read back with `small`)
                            pending.append(rb)
                    else:
                        self.getRuleBook(req[1], req[2], req[3], req[4], _defer=True)
                except NotImplementedError:
                    break  # not buildable for THIS batch (e.g. normals now guide a strided scale): the op itself will say so
            if pending or self._grid_status or batch_small is not None:
                # ONE device->host copy for the pair counts of all submanifold rulebooks, for the site counts the
                # grids built without read-back found themselves, and for the "gave up" words of the stream-K control blocks
                health = _sk_health_words(c.device)
                parts = [rb.counts.sum().reshape(1) for rb in pending] + [s for s, _ in self._grid_status] + health
                if batch_small is not None:
                    parts.append(batch_small[0])
                flat = parts[0] if len(parts) == 1 else torch.cat(parts)
                n_health, dev_ = len(health), c.device

                def finish(totals, self=self, pending=pending, batch_small=batch_small):
                    if batch_small is not None:
                        self._batch_finish(totals[len(totals) - batch_small[0].numel():], batch_small[1])
                        totals = totals[:len(totals) - batch_small[0].numel()]
                    for rb, t in zip(pending, totals):
                        rb.total_pairs = int(t)
                    self._check_grid_status(totals[len(pending):len(totals) - n_health])
                    if n_health:
                        _sk_check(totals[len(totals) - n_health:], dev_)

                if in_prefetch and batch_small is not None and DEFER_PREFETCH_READBACK:
                    # a prefetch whose launches are all queued (site counts known from the chain): the read-back is an
                    # asynchronous copy into pinned memory, picked up when an InputLayer call adopts the batch -- the host
                    # thread of a serving loop goes on to issue the current forward pass instead of waiting for the side stream
                    host = _pinned(flat.numel(), flat.dtype)
                    host[:flat.numel()].copy_(flat, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    self._deferred = (host, flat.numel(), ev, finish)
                else:
                    finish(flat.cpu().tolist())
        elif batch_small is not None:  # (not reachable: the batch build needs a recorded plan; kept for safety)
            self._batch_finish(batch_small[0].cpu().tolist(), batch_small[1])

    def _chain(self, c, key, old_plan, occ=False, hashed=True):
        """Site counts of every scale the previous forward pass on this input size built (its strided rulebook
        requests chain down from the input size): from the occupancy bitmaps of the batch's bounding box (scn_occ_bounds +
        scn_occ_chain: two read-backs, and the bitmaps are what scn_batch_build_occ then builds everything from), or with ONE
        read-back from hash sets (scn_grid_chain_counts)."""
        sizes = [key]
        for req in old_plan:
            if req[0] != "sub" and req[1] == sizes[-1] and req[3] == (2, 2, 2) and req[4] == (2, 2, 2):
                sizes.append(req[2])
        self._occ = None
        if not old_plan or len(sizes) > _GRID_CHAIN_MAX:
            return None, sizes
        lib = _lib_()
        P = c.size(0)
        if P <= 0 or P >= (1 << 30):  # (scn_grid_chain_counts takes fewer points than scn_grid_build: per-scale read-backs then)
            return None, sizes
        if occ:  # (also for a plan of ONE scale -- a network without strided convolutions --, which the hash-set chain leaves alone)
            mm = (ctypes.c_int32 * 8)()
            scratch = torch.empty(16, dtype=torch.int32, device=c.device)
            check(lib.scn_occ_bounds(ptr(c), _COORD_DTYPES[c.dtype], c.size(1), P, ptr(scratch), mm, stream()))
            box = _lib.OccBox()
            check(lib.scn_occ_plan(mm, len(sizes), ctypes.byref(box)))
            if box.bytes > 0 and (occ is True or box.woff[len(sizes)] <= max(1 << 20, OCC_WORDS_PER_ROW * P)):
                buf = torch.empty(box.bytes, dtype=torch.uint8, device=c.device)
                check(lib.scn_occ_chain(ptr(c), _COORD_DTYPES[c.dtype], c.size(1), P, ctypes.byref(box), ptr(buf), box.bytes,
                                        None, stream()))
                self._occ = (box, buf)

                def fetch_occ():
                    n = (ctypes.c_int32 * len(sizes))()
                    check(lib.scn_occ_fetch(ptr(buf), ctypes.byref(box), n, stream()))
                    self._chain_counts = {sz: int(v) for sz, v in zip(sizes, n)}
                    return self._chain_counts
                return fetch_occ, sizes
            OCC_STATS["too_large"] += 1
        if not hashed or len(sizes) < 2:
            return None, sizes
        tb = lib.scn_grid_chain_temp_bytes(P, len(sizes))
        if tb == 0:
            return None, sizes
        temp = torch.empty(tb, dtype=torch.uint8, device=c.device)
        ev = _prof_begin("rules")
        check(lib.scn_grid_chain_counts(ptr(c), _COORD_DTYPES[c.dtype], c.size(1), P, len(sizes), ptr(temp), tb, None,
                                        stream()))  # (queued; the read-back comes when the first grid's launches are queued too)
        _prof_end(ev, "grid_chain", 0.0, float(P * c.size(1) * c.element_size() + tb), (P, len(sizes)))

        def fetch():
            n = (ctypes.c_int32 * len(sizes))()
            check(lib.scn_grid_chain_fetch(ptr(temp), P, len(sizes), n, stream()))
            self._chain_counts = {sz: int(v) for sz, v in zip(sizes, n)}
            return self._chain_counts
        return fetch, sizes

    def _batch_build(self, c, sizes, counts, old_plan, skip_sub=()):
        """scn_batch_build: the grids of the chain `sizes` (site counts known: `counts`), the 2/2 rulebooks between them and the
        submanifold rulebooks the previous forward pass asked for, queued by one call into one arena; every array the ops read
        is a view into it.  Returns (the `small` block to read back, what _batch_finish needs)."""
        lib = _lib_()
        dev = c.device
        L = len(sizes)
        plan = _lib.BatchPlan()
        plan.levels, plan.coords_dtype, plan.ncols, plan.P = L, _COORD_DTYPES[c.dtype], c.size(1), c.size(0)
        want_sub = {req[1] for req in old_plan if req[0] == "sub" and req[2] == (3, 3, 3)}
        for l, sz in enumerate(sizes):
            plan.N[l] = counts[sz]
            plan.sub[l] = 1 if sz in want_sub and sz not in skip_sub else 0
            plan.strided[l] = 1 if l + 1 < L else 0
        lo = _lib.BatchLayout()
        plan.flags = 1 if self._occ is not None else 0
        check(lib.scn_batch_layout_query(ctypes.byref(plan), ctypes.byref(lo)))
        if self._occ is not None:
            # results and scratch apart: the results live as long as the batch, the scratch (the neighbour tables of all scales, the
            # input rows' arrival numbers: ~110 bytes per site) and the bitmaps go back to the allocator behind this call -- it is
            # stream-ordered, and the stream has the launches that use them (round-5 advisor: one allocation kept the scratch alive
            # with the batch, twice over while a prefetched batch waits)
            box, buf = self._occ
            arena = torch.empty(lo.scratch, dtype=torch.uint8, device=dev)
            scratch = torch.empty(lo.total - lo.scratch, dtype=torch.uint8, device=dev)
            check(lib.scn_batch_build_occ(ptr(c), ctypes.byref(plan), ctypes.byref(box), ptr(buf), ptr(arena), lo.scratch,
                                          ptr(scratch), lo.total - lo.scratch, stream()))
            self._occ = None
            del scratch, buf
            OCC_STATS["built"] += 1
        else:
            arena = torch.empty(lo.total, dtype=torch.uint8, device=dev)
            check(lib.scn_batch_build(ptr(c), ctypes.byref(plan), ptr(arena), lo.total, stream()))
        a32, a64 = arena.view(torch.int32), arena.view(torch.int64)

        def i32(off, *shape):
            st, acc = [], 1
            for d in reversed(shape):
                st.append(acc)
                acc *= d
            return a32.as_strided(shape, tuple(reversed(st)), off >> 2)

        def i64(off, n):
            return a64.as_strided((n,), (1,), off >> 3)

        P = c.size(0)
        small = i32(lo.small, L * 37)
        self.input_order, self.input_site_of_row = i32(lo.order, P), i32(lo.site_of_row, P)
        self.input_site_start = i32(lo.site_start, plan.N[0] + 1)
        subs = []
        for l, sz in enumerate(sizes):
            N = int(plan.N[l])
            g = _Grid(N, i32(lo.coords[l], N, 4), i64(lo.keys[l], N))
            self.grids[sz] = g
            self.nActive[sz] = N
            n_gran = (N + GRANULE - 1) // GRANULE
            if plan.sub[l]:
                cap = int(lo.hash_cap[l])
                if cap:  # (a build from occupancy bitmaps has no hash maps: _hash() makes one when something asks)
                    g.hash_keys, g.hash_vals, g.hash_cap = i64(lo.hash_keys[l], cap), i32(lo.hash_vals[l], cap), cap
                rb = _RuleBook(27, N, i32(lo.sub_rules[l], 27, N, 2), small[2 * L + 27 * l:2 * L + 27 * l + 27])
                rb.bases[1] = i32(lo.sub_bases[l], 27, n_gran + 1)
                rb.n_rows[1] = N
                if STREAM_K and (27, 1) in _SK_WANTED and (N + 127) // 128 >= _SK_MIN_TILES:
                    _tile_passes(rb, 1, N)
                self.submanifoldRuleBooks[sz + (3, 3, 3)] = rb
                subs.append((l, rb))
            if plan.strided[l]:
                Nc = int(plan.N[l + 1])
                n_gran_c = (Nc + GRANULE - 1) // GRANULE
                o = 2 * L + 27 * L + 8 * l
                rb = _RuleBook(8, N, i32(lo.str_rules[l], 8, N, 2), small[o:o + 8])
                rb.bases[1], rb.n_rows[1] = i32(lo.str_bases_coarse[l], 8, n_gran_c + 1), Nc
                rb.bases[0], rb.n_rows[0] = i32(lo.str_bases_fine[l], 8, n_gran + 1), N
                rb.total_pairs = N  # every fine site appears in exactly one list
                rb.unique_col = 0
                if STREAM_K:
                    for col, n_out in ((1, Nc), (0, N)):
                        if (8, col) in _SK_WANTED and (n_out + 127) // 128 >= _SK_MIN_TILES:
                            _tile_passes(rb, col, n_out)
                self.ruleBooks[sz + (2, 2, 2, 2, 2, 2)] = rb
        return small, (L, [int(plan.N[l]) for l in range(L)], subs)

    def _batch_finish(self, small, info):
        """the host's half of a batch build: `small` read back -- [levels][2] grid status, [levels][27] and [levels][8] list lengths"""
        L, want, subs = info
        for l in range(L):
            got, err = int(small[2 * l]), int(small[2 * l + 1])
            if err == 2:
                # scn_batch_build_occ: a voxel holds more than SCN_OCC_SEGMENT_MAX input rows and its rows in `order` are in the
                # order the atomics handed out.  Rare (a degenerate cloud): sort (site, row) here.  The build has finished (this
                # runs behind its read-back); the copy is queued on the current stream, which consumes `order` from here on.
                P = self.input_P
                k = (self.input_site_of_row.to(torch.int64) << 32) | torch.arange(P, dtype=torch.int64, device=self.input_order.device)
                self.input_order.copy_((torch.sort(k).values & 0xFFFFFFFF).to(torch.int32))
                OCC_STATS["order_repaired"] += 1
                err = 0
            if err:
                raise RuntimeError("libscn_hip: scn_batch_build: coordinate outside the 31-bit key box (0<=x<1024, 0<=y<2048, "
                                   "0<=z<1024) or bad sample index")
            if got != want[l]:
                raise RuntimeError("occuseg_amd: grid of scale %d built for %d sites found %d: scn_grid_chain_counts and "
                                   "scn_batch_build disagree" % (l, want[l], got))
        for l, rb in subs:
            rb.total_pairs = int(sum(small[2 * L + 27 * l:2 * L + 27 * l + 27]))

    def _flush_grid_status(self):
        if self._grid_status:  # (a grid built outside the replayed plan: its own read-back)
            self._check_grid_status(torch.cat([s for s, _ in self._grid_status]).cpu().tolist())

    def _check_grid_status(self, flat):
        """[N found, error flag] of every grid that was built for a site count taken from the chain"""
        st, self._grid_status = self._grid_status, []
        for i, (_, want) in enumerate(st):
            got, err = int(flat[2 * i]), int(flat[2 * i + 1])
            if err or got != want:
                raise RuntimeError("occuseg_amd: grid built for %d sites found %d (error flag %d): "
                                   "scn_grid_chain_counts and scn_grid_build disagree" % (want, got, err))

    def _tensors(self):
        for g in self.grids.values():
            for t in (g.coords, g.keys, g.hash_keys, g.hash_vals):
                if t is not None:
                    yield t
        for rb in list(self.submanifoldRuleBooks.values()) + list(self.ruleBooks.values()):
            for t in [rb.rules, rb.counts, rb.nbr, rb.ori] + list(rb.bases.values()) + list(rb.passes.values()):
                if t is not None:
                    yield t
        for sn, ori in self.normals.values():
            yield sn
            yield ori
        for t in (self.input_order, self.input_site_of_row, self.input_site_start):
            if t is not None:
                yield t

    def _complete(self):
        """the deferred half of a prefetched build: what its read-back says (see inputLayer)"""
        d, self._deferred = self._deferred, None
        if d is not None:
            host, n, ev, finish = d
            ev.synchronize()
            vals = host[:n].tolist()
            _PINNED.append(host)
            finish(vals)

    def _adopt(self, other, done):
        """take over a Metadata_3 that prefetch_input built on the side stream"""
        other._complete()
        scale = self.normal_guide_scale
        self.__dict__.update(other.__dict__)
        self.normal_guide_scale = scale
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        for t in self._tensors():  # allocated on the side stream, consumed on this one
            t.record_stream(cur)
        _PLANS[self.inputSpatialSize] = self._plan

    def _record(self, req):
        if req not in self._plan:
            self._plan.append(req)
        _PLANS_USED[self.inputSpatialSize] = self._plan

    def getNActive(self, spatialSize):
        return self.nActive[_size_key(spatialSize)]

    def getSpatialLocations(self, spatialSize):
        """LongTensor [N,4] = (x,y,z,batch) in site order (sparseconvnet/SCN/Metadata/Metadata.cpp:722-749)."""
        return self.grids[_size_key(spatialSize)].coords.cpu().to(torch.int64)

    # -- rulebooks -----------------------------------------------------------
    def _tile_bases(self, rb, out_col, n_out):
        lib = _lib_()
        n_gran = (n_out + GRANULE - 1) // GRANULE
        bases = torch.empty((rb.K, n_gran + 1), dtype=torch.int32, device=rb.rules.device)
        check(lib.scn_tile_bases(ptr(rb.rules), ptr(rb.counts), rb.K, rb.stride, out_col, n_out,
                                 ptr(bases), stream()))
        rb.bases[out_col] = bases
        rb.n_rows[out_col] = n_out
        # pass prefix of the 128-row tiles (scn_conv_forward's tile_passes): what a stream-K launch cuts into equal slices.
        # Three small launches per rulebook column: built with the rulebook only when a launch of the big gather-mm kernel has
        # asked for the tables of such a rulebook before (_SK_WANTED; the first time _conv builds them itself) -- a network
        # whose planes keep it off that kernel (the m32 UNet of the forward-only configuration) never pays for them.
        if STREAM_K and (rb.K, out_col) in _SK_WANTED and (n_out + 127) // 128 >= _SK_MIN_TILES:
            _tile_passes(rb, out_col, n_out)

    def getSubmanifoldRuleBook(self, spatialSize, filterSize, dilated_rate=1, _defer=False):
        """cache key (spatialSize, filterSize) as sparseconvnet/SCN/Metadata/Metadata.cpp:504-529"""
        fs = _size_key(filterSize)
        if fs != (3, 3, 3):
            raise NotImplementedError("SubmanifoldConvolution: the reference GPU rule builder aborts unless "
                                      "filter_size == 3 (sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cpp:40-44)")
        if int(dilated_rate) != 1:
            raise NotImplementedError("SubmanifoldConvolution: dilated_rate must be 1 (the reference GPU "
                                      "builder ignores it, SubmanifoldConvolutionRules.h:248-275)")
        key = _size_key(spatialSize) + fs
        rb = self.submanifoldRuleBooks.get(key)
        if rb is None:
            lib = _lib_()
            g = self._hash(self.grids[_size_key(spatialSize)])
            dev = g.coords.device
            N = g.n
            nbr = torch.empty((27, N), dtype=torch.int32, device=dev)
            rules = torch.empty((27, N, 2), dtype=torch.int32, device=dev)
            counts = torch.empty(27, dtype=torch.int32, device=dev)
            tb = lib.scn_rules_temp_bytes(N, 27)
            temp = torch.empty(tb, dtype=torch.uint8, device=dev)
            # normals known at this scale (Metadata::getSubmanifoldRuleBook, sparseconvnet/SCN/Metadata/Metadata.cpp:514-530):
            # remap every pair to list rot_index[class of its output site][offset]
            ori = self.normals[key[:3]][1] if key[:3] in self.normals else None
            nbr_tmp = torch.empty((27, N), dtype=torch.int32, device=dev) if ori is not None else None
            ev = _prof_begin("rules")
            check(lib.scn_submanifold_rules(ptr(g.coords), N, ptr(g.hash_keys), ptr(g.hash_vals), g.hash_cap,
                                            ptr(nbr), ptr(rules), ptr(counts), ptr(temp), tb, ptr(ori),
                                            1 if ori is not None else 0, -1, ptr(nbr_tmp), stream()))
            rb = _RuleBook(27, N, rules, counts)
            rb.ori = ori
            # read 16 N (coords), write + read the 27 probe results (2 x 108 N), write 8 R
            _prof_end(ev, "sub_rules", 0.0, lambda rb=rb, N=N: 16.0 * N + 216.0 * N + 8.0 * (rb.total_pairs or 0), (N,))
            if KEEP_NBR:  # the probe table (108 B/site) is scratch of the build; tests ask to see it
                rb.nbr = nbr
            self._tile_bases(rb, 1, N)
            if not _defer:
                rb.total_pairs = int(counts.sum().item())
            self.submanifoldRuleBooks[key] = rb
        if not _defer:
            if rb.total_pairs is None:
                rb.total_pairs = int(rb.counts.sum().item())
            self._record(("sub", key[:3], fs))
        return rb

    def _class_books(self, spatialSize, rb):
        """the three plain rulebooks {outputs of class c} a normal-guided rulebook's dInput runs over (see
        SubmanifoldConvolution_backward); built once per batch, on first use"""
        if rb.class_books is None:
            lib = _lib_()
            g = self._hash(self.grids[_size_key(spatialSize)])
            dev, N = g.coords.device, g.n
            books = {}
            for cls in (0, 2, 4):
                nbr = torch.empty((27, N), dtype=torch.int32, device=dev)
                rules = torch.empty((27, N, 2), dtype=torch.int32, device=dev)
                counts = torch.empty(27, dtype=torch.int32, device=dev)
                tb = lib.scn_rules_temp_bytes(N, 27)
                temp = torch.empty(tb, dtype=torch.uint8, device=dev)
                check(lib.scn_submanifold_rules(ptr(g.coords), N, ptr(g.hash_keys), ptr(g.hash_vals), g.hash_cap,
                                                ptr(nbr), ptr(rules), ptr(counts), ptr(temp), tb, ptr(rb.ori), 2, cls,
                                                None, stream()))
                b = _RuleBook(27, N, rules, counts)
                self._tile_bases(b, 0, N)  # for a fixed offset the input site is unique and ascending with the output site
                books[cls] = b
            totals = torch.stack([b.counts.sum() for b in books.values()]).cpu().tolist()
            for b, t in zip(books.values(), totals):
                b.total_pairs = int(t)
            rb.class_books = books
        return rb.class_books

    def getRuleBook(self, inputSpatialSize, outputSpatialSize, size, stride, _defer=False):
        """Metadata::getRuleBook (sparseconvnet/SCN/Metadata/Metadata.cpp:598-625): builds the coarse
        grid as a side effect and sets nActive[outputSpatialSize]."""
        fs, st = _size_key(size), _size_key(stride)
        if fs != (2, 2, 2) or st != (2, 2, 2):
            raise NotImplementedError("Convolution/Deconvolution: the reference GPU path asserts size == stride == 2 "
                                      "(sparseconvnet/SCN/Metadata/ConvolutionRules.h:353-358)")
        iS, oS = _size_key(inputSpatialSize), _size_key(outputSpatialSize)
        key = iS + fs + st
        rb = self.ruleBooks.get(key)
        if rb is None and iS in self.normals and iS[0] >= self.normal_guide_scale:
            # The reference's GPU builder for this case cannot have run: in its rule loop the output index is consumed
            # TWICE per (input, output) pair -- `out_index[query_index++]` for the orientation and again for the pair
            # (sparseconvnet/SCN/Metadata/ConvolutionRules.h:225,228) -- while `out_index` holds ONE entry per pair
            # (:164-178): the orientation is taken from another site's normal and from pair n/2 on the reads run past the
            # end of the vector (undefined behaviour).  There is no reference behaviour to reproduce; the fork's shipped
            # configuration never gets here (rotation_guide_level 0: examples/ScanNet/config.py:50, model.py:626).
            raise NotImplementedError(
                "normal-guided STRIDED rules (input spatial size %d >= normal_guide_scale %d) are not built: the "
                "reference's own GPU path for them (sparseconvnet/SCN/Metadata/ConvolutionRules.h:139-236) indexes "
                "out of bounds (query_index is incremented twice per pair, :225 and :228) and has no defined result; with "
                "the default rotation_guide_level 0 (examples/ScanNet/config.py:50, model.py:626) the normals guide the "
                "input scale only, which is what this backend provides" % (iS[0], self.normal_guide_scale))
        if rb is None:
            lib = _lib_()
            fine = self.grids[iS]
            dev = fine.coords.device
            Nf = fine.n
            coarse, _order, coarse_of_fine, _start = self._build_grid(fine.coords, 0, 4, 1, Nf,
                                                                      n_known=self._chain_counts.get(oS))
            self.grids[oS] = coarse
            self.nActive[oS] = coarse.n
            rules = torch.empty((8, Nf, 2), dtype=torch.int32, device=dev)
            counts = torch.empty(8, dtype=torch.int32, device=dev)
            tb = lib.scn_rules_temp_bytes(Nf, 8)
            temp = torch.empty(tb, dtype=torch.uint8, device=dev)
            ev = _prof_begin("rules")
            check(lib.scn_strided_rules(ptr(fine.coords), Nf, ptr(coarse_of_fine), ptr(rules), ptr(counts),
                                        ptr(temp), tb, stream()))
            _prof_end(ev, "strided_rules", 0.0, 2.0 * (16.0 + 4.0) * Nf + 8.0 * Nf, (Nf,))
            rb = _RuleBook(8, Nf, rules, counts)
            self._tile_bases(rb, 1, coarse.n)  # lists are also ascending in the coarse id
            self._tile_bases(rb, 0, Nf)
            rb.total_pairs = Nf  # every fine site appears in exactly one list
            rb.unique_col = 0
            self.ruleBooks[key] = rb
        if not _defer:
            self._record(("conv", iS, oS, fs, st))
            self._flush_grid_status()
        return rb

    # -- host views for tests / tooling ---------------------------------------
    def getInputRulesHost(self):
        """reference-format input rulebook: (header [mode,maxActive,P,N], table [N,1+maxActive])"""
        lib = _lib_()
        N = self.nActive[self.inputSpatialSize]
        scratch = torch.zeros(1, dtype=torch.int32, device=self.device)
        ma = ctypes.c_int32(0)
        check(lib.scn_input_rules_max_active(ptr(self.input_site_start), N, ptr(scratch), ctypes.byref(ma), stream()))
        table = torch.empty((N, 1 + ma.value), dtype=torch.int32, device=self.device)
        check(lib.scn_input_rules_table(ptr(self.input_order), ptr(self.input_site_start), N, ma.value,
                                        ptr(table), stream()))
        header = np.array([self.input_mode, ma.value, self.input_P, N], np.int32)
        return header, table.cpu().numpy()


def _norm_device(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


PREFETCH_AT_COARSE_LEVELS = os.environ.get("SCN_PREFETCH_AT_COARSE", "1") != "0"  # A/B switch, see prefetch_input
_COARSE_MARK = {}        # device index -> [event, input-layer count at which it was recorded]
_STEPS = {}              # device index -> InputLayer calls that built or adopted a batch there (prefetches not counted).  Per
                         # device, and "this call is a prefetch" travels as an argument of inputLayer: as process-global flags
                         # (round 4) one DataParallel-style thread's prefetch flipped the mode of another thread's InputLayer call


def _step_begins(device):
    _STEPS[device.index] = _STEPS.get(device.index, 0) + 1


def _step_count(device):
    return _STEPS.get(device.index, 0)


PREFETCH_PRIORITY = int(os.environ.get("SCN_PREFETCH_PRIORITY", "-1"))  # HIP stream priority of the rule-build side stream (-1 = high)


def prefetch_input(spatialSize, coords, mode, device, wait_for_current_stream=True, normals=None,
                   normal_guide_scale=None):
    """Build the grids and (replaying the previous forward pass's requests) the rulebooks that the next
    ``InputLayer`` call on exactly this ``coords`` tensor will need, on a high-priority side stream, so that
    the build and its data-dependent host synchronisations overlap the current step's queued kernels.  The
    InputLayer call then adopts the result instead of building (it waits on an event, not on the host).
    ``coords`` must not be modified in between.  ``wait_for_current_stream=False`` may be passed when a CUDA
    ``coords`` tensor was produced long ago (otherwise the side stream first waits for the current one, which
    serialises the build behind the queued step)."""
    device = _norm_device(device)
    assert coords.dtype in _COORD_DTYPES and coords.dim() == 2
    side = _SIDE_STREAMS.get((device, PREFETCH_PRIORITY))
    if side is None:
        side = _SIDE_STREAMS[(device, PREFETCH_PRIORITY)] = torch.cuda.Stream(device=device, priority=PREFETCH_PRIORITY)
    if coords.is_cuda and wait_for_current_stream:
        side.wait_stream(torch.cuda.current_stream(device))
    m = Metadata_3()
    if normals is not None and normals.is_cuda and wait_for_current_stream:
        side.wait_stream(torch.cuda.current_stream(device))
    if normal_guide_scale is not None:  # the threshold getRuleBook tests must be the InputLayer's, not the default
        m.setNormalGuideScale(normal_guide_scale)
    # WHERE in the current step the build runs (round 4): a training loop issues the prefetch when the whole step is queued -- the
    # GPU is then half a step behind, and the build's ~60 small kernels landed beside the first big dInput launches of the
    # backward pass (0.5 ms per step: they are vector-ALU work on SIMDs that run MFMAs).  The backward pass marks the point
    # where it enters the coarse levels -- launches of a few hundred workgroups that leave most of the chip's issue slots
    # idle -- and the build waits for that mark.
    mark = _COARSE_MARK.get(device.index) if PREFETCH_AT_COARSE_LEVELS else None
    if mark is not None and mark[1] == _step_count(device):
        side.wait_event(mark[0])
    with torch.cuda.stream(side):
        m.inputLayer(spatialSize, coords, 0, mode, device, normals=normals, _in_prefetch=True)
        done = torch.cuda.Event()
        done.record(side)
    _PREFETCHED.clear()  # one batch ahead is all a training loop needs; never keep stale grids alive
    _PREFETCHED[id(coords)] = (coords, (_size_key(spatialSize), int(mode), device, m.normal_guide_scale), m, done, normals)


# ----------------------------------------------------------------- helpers ---
# Weights re-laid out in MFMA fragment order (scn_weight_pack), one packed copy per (weight tensor, transposed, reversed).
# The reference reads the module's weight tensor on every call (sparseconvnet/SCN/CUDA/Convolution.cpp:105-162); a packed
# copy is therefore only ever used within one WEIGHT EPOCH and one value of the tensor's version counter.  The version
# counter alone is NOT enough: torch's fused optimizers and writes through `.data` do not move it (round-2 verdict: the
# convolutions kept computing with first-step weights).  The epoch moves
#   * at every InputLayer call (a new batch: once per forward pass),
#   * at the first forward-direction call after any backward call (a loop that re-uses one input tensor),
#   * after every torch optimizer step (global post-step hook) and in FlatGradBucket.zero(),
#   * when invalidate_weight_cache() is called (writes through `.data` between two calls on the same input).
# The first lookup in a new epoch re-packs ALL live entries -- both orientations of every layer -- in ONE launch
# (scn_weight_pack_batch: 0.7 GB of traffic for the m64 UNet instead of 74 launches).  WPACK_CACHE = False: no cache at
# all, scn_conv_forward packs on every call.
import weakref  # noqa: E402

_WPACK = {}          # key -> _PackedWeight
_WEPOCH = [1]
_BWD_SEEN = [False]
_WTABLE = {}         # device index -> (signature, descriptor tensor, total blocks)
_REPACK_EVENT = {}   # device index -> event recorded behind the last batched re-pack (for readers on other streams)
WPACK_CACHE = True
WPACK_STATS = {"single": 0, "batch": 0, "batch_entries": 0}  # launches since import (tests, bench)


class _PackedWeight(object):
    __slots__ = ("ref", "off", "ptr", "version", "epoch", "wpack", "dims", "flags", "blocks")


def weights_changed():
    """start a new weight epoch: every packed copy is re-made from the module's weights before its next use"""
    _WEPOCH[0] += 1


# Forward scopes (round 6): every container of the operator API brackets its forward with forward_enter / forward_exit, and the
# OUTERMOST one starts a new weight epoch -- whatever was written into a weight tensor since the last forward sweep, by any
# means (a write through `.data` moves no version counter and fires no hook), is seen by the next sweep, as in the reference,
# which reads the module's weights on every call (sparseconvnet/SCN/CUDA/Convolution.cpp:105-162).  A convolution called
# directly, outside any container, re-packs its own weights (weight_touched).  Cost: none beyond what a forward pass already
# paid -- the epoch moved at its InputLayer call anyway, and epochs that move with no lookup in between cost nothing.
_FWD_DEPTH = [0]


def forward_enter():
    if _FWD_DEPTH[0] == 0:
        _WEPOCH[0] += 1
    _FWD_DEPTH[0] += 1


def forward_exit():
    _FWD_DEPTH[0] -= 1


def weight_touched(weight):
    """the packed copies of THIS weight tensor are re-made at their next use (one small launch each)"""
    base = weight._base if weight._base is not None else weight
    for t in (False, True):
        for r in (False, True):
            ent = _WPACK.get((id(base), weight.data_ptr(), t, r))
            if ent is not None:
                ent.version = -1


def invalidate_weight_cache():
    _WPACK.clear()
    _WTABLE.clear()
    _WEPOCH[0] += 1


def _optimizer_step_hook(_opt, _args, _kwargs):
    _WEPOCH[0] += 1


try:  # global: fires after the step of EVERY torch optimizer (fused ones included)
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_hook
    _reg_hook(_optimizer_step_hook)
except ImportError:  # pragma: no cover  (older torch: InputLayer / backward / FlatGradBucket.zero() still move the epoch)
    pass


def _repack_all(device):
    """one launch re-packs every live entry on `device` and stamps it with the current epoch"""
    lib = _lib_()
    live = []
    for key, ent in list(_WPACK.items()):
        base = ent.ref()
        if base is None or base.data_ptr() + ent.off != ent.ptr:
            _WPACK.pop(key, None)  # freed, or re-seated through `.data = ...`: the next use packs afresh
            continue
        if base.device == device:
            if ent.blocks > 0:
                live.append((ent, base))
            else:  # planes that are not multiples of 16: the generic kernel reads the module's weights itself
                ent.version, ent.epoch = base._version, _WEPOCH[0]
    if not live:
        return
    sig = tuple((ent.ptr, ent.wpack.data_ptr()) + ent.dims + (ent.flags,) for ent, _ in live)
    tab = _WTABLE.get(device.index)
    if tab is None or tab[0] != sig:
        desc = np.zeros(len(live), dtype=np.dtype([("W", "<u8"), ("Wp", "<u8"), ("K", "<i4"), ("Ci", "<i4"), ("Co", "<i4"),
                                                   ("flags", "<i4"), ("first", "<i8")]))
        first = 0
        for i, (ent, _) in enumerate(live):
            desc[i] = (ent.ptr, ent.wpack.data_ptr(), ent.dims[0], ent.dims[1], ent.dims[2], ent.flags, first)
            first += ent.blocks
        dev_desc = torch.from_numpy(desc.view(np.uint8).copy()).to(device)
        tab = _WTABLE[device.index] = (sig, dev_desc, first)
    check(lib.scn_weight_pack_batch(ptr(tab[1]), len(live), tab[2], stream()))
    # (the packed copies are re-made on the CURRENT stream; a reader on another stream -- DataParallel-style threads -- must
    #  wait for this: _REPACK_EVENT[device index] is recorded behind every batched re-pack)
    ev = _REPACK_EVENT.get(device.index)
    if ev is None:
        ev = _REPACK_EVENT[device.index] = [torch.cuda.Event(), 0]
    cur = torch.cuda.current_stream(device)
    ev[0].record(cur)
    ev[1] = cur.cuda_stream
    WPACK_STATS["batch"] += 1
    WPACK_STATS["batch_entries"] += len(live)
    for ent, base in live:
        ent.version = base._version
        ent.epoch = _WEPOCH[0]


def _packed_weight(weight, w_transposed, w_reverse, K, Ci, Co, stream_handle=None):
    """returns (packed tensor, ready flag for scn_conv_forward)"""
    if not w_transposed and _BWD_SEEN[0]:  # a forward-direction call after a backward call: weights may have been updated
        _BWD_SEEN[0] = False
        _WEPOCH[0] += 1
    if not WPACK_CACHE:
        return torch.empty(K * Ci * Co, dtype=torch.float32, device=weight.device), 0
    base = weight._base if weight._base is not None else weight
    key = (id(base), weight.data_ptr(), bool(w_transposed), bool(w_reverse))
    ent = _WPACK.get(key)
    if ent is not None and (ent.ref() is not base or ent.dims != (K, Ci, Co) or ent.ptr != weight.data_ptr()):
        ent = None
    if ent is not None and ent.epoch != _WEPOCH[0]:
        _repack_all(weight.device)  # stamps every live entry, this one included (unless it was dropped)
        ent = _WPACK.get(key)
    if ent is not None and ent.epoch == _WEPOCH[0] and ent.version == weight._version:
        ev = _REPACK_EVENT.get(weight.device.index)
        if ev is not None and ev[1] != (stream() if stream_handle is None else stream_handle):  # re-packed on ANOTHER stream: wait for that launch
            torch.cuda.current_stream(weight.device).wait_event(ev[0])
        return ent.wpack, 1
    lib = _lib_()
    if ent is None:
        ent = _PackedWeight()
        ent.wpack = torch.empty(K * Ci * Co, dtype=torch.float32, device=weight.device)

        def _drop(_ref, key=key):
            _WPACK.pop(key, None)
        ent.ref = weakref.ref(base, _drop)
        ent.ptr = weight.data_ptr()
        ent.off = ent.ptr - base.data_ptr()
        ent.dims = (K, Ci, Co)
        ent.flags = lib.scn_weight_pack_flags(int(w_transposed), int(w_reverse), K, Ci, Co)
        ent.blocks = lib.scn_weight_pack_blocks(K, Ci, Co)
        _WPACK[key] = ent
    if ent.blocks > 0:
        check(lib.scn_weight_pack(ptr(weight), int(w_transposed), int(w_reverse), K, Ci, Co, ptr(ent.wpack), stream()))
        WPACK_STATS["single"] += 1
    ent.version = weight._version
    ent.epoch = _WEPOCH[0]
    return ent.wpack, 1


class BnLink(object):
    """What a batch norm leaves for the convolution that consumes its output (not in the reference).  In the backward pass
    that convolution's dInput launch IS the batch norm's d_output_features: it takes the two reductions of the batch norm's
    backward pass from its output tile (scn_conv_forward's bn_bwd) and leaves them here, keyed on the gradient tensor it
    returned; BatchNormalization_backward then skips its statistics pass over x and d_out."""
    __slots__ = ("x", "save_mean", "save_invstd", "weight", "bias", "leakiness", "partials", "grad_key")

    def __init__(self, x, save_mean, save_invstd, weight, bias, leakiness):
        self.x, self.save_mean, self.save_invstd = x, save_mean, save_invstd
        self.weight = weight if weight is not None and weight.numel() else None
        self.bias = bias if bias is not None and bias.numel() else None
        self.leakiness = float(leakiness)
        self.partials, self.grad_key = None, None

    def struct(self):
        return _lib.BnBwdStats(ptr(self.x), ptr(self.save_mean), ptr(self.save_invstd), ptr(self.weight), ptr(self.bias),
                               self.leakiness)

    def offer(self, grad, partials):
        self.partials = partials
        # (a weak reference, not the address: a hook that clones and frees the gradient can put another tensor at that address)
        self.grad_key = (weakref.ref(grad), grad._version) if partials is not None else None

    def take(self, grad):
        """the partial sums, if `grad` is exactly the tensor they were taken from (nothing was added into it since)"""
        p, key = self.partials, self.grad_key
        self.partials, self.grad_key = None, None
        if p is not None and grad.is_contiguous() and key[0]() is grad and key[1] == grad._version:
            return p
        return None


BN_BWD_STATS_FROM_CONV = True  # A/B switch


def _row_slice(t):
    """2-D, unit stride along the planes, 16-byte aligned rows: a column slice of a wider row-major float tensor"""
    return (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.size(1) and t.stride(0) % 4 == 0
            and t.data_ptr() % 16 == 0 and t.size(0) > 0)


_PLAN_CACHE = {}
_PLAN_CACHEABLE = {}


def _plan(lib, K, stride, n_plan, n_out, Ci, Co, uniq):
    """(scn_conv_plan, scn_conv_workspace_bytes, scn_conv_stats_blocks) of a launch shape, asked once per shape and library build:
    three ctypes calls per launch are host time the forward-only configuration is bound by"""
    key = (id(lib), K, stride, n_plan, n_out, Ci, Co, uniq)
    r = _PLAN_CACHE.get(key)
    if r is None:
        r = (lib.scn_conv_plan(K, stride, n_plan, n_out, Ci, Co, uniq),
             lib.scn_conv_workspace_bytes(K, stride, n_plan, n_out, Ci, Co, uniq),
             lib.scn_conv_stats_blocks(K, stride, n_plan, n_out, Ci, Co, uniq))
        cacheable = _PLAN_CACHEABLE.get(id(lib))
        if cacheable is None:  # (the tuning library's answers change with scn_set_tuning: never cached)
            cacheable = _PLAN_CACHEABLE[id(lib)] = not hasattr(lib, "scn_set_tuning")
        if cacheable:
            if len(_PLAN_CACHE) > 4096:
                _PLAN_CACHE.clear()
            _PLAN_CACHE[key] = r
    return r


def _conv(inp, out, weight, bias, rb, in_col, out_col, n_out, w_transposed=False, w_reverse=False, residual=None,
          stats_out=None, bn_link=None, out_view=None):
    """out = gather-mm-scatter over rb; weight is the module's [K, a, b] tensor, used as [K, Ci, Co]
    (forward) or, with w_transposed, as its per-offset transpose (dInput).  When every output row appears in exactly
    one pair of the rulebook (rb.unique_col == out_col: the fine column of a strided rulebook, the identity list)
    the backend runs the launch without accumulating across offsets."""
    lib = _lib_()
    K = weight.size(0)
    Ci, Co = (weight.size(2), weight.size(1)) if w_transposed else (weight.size(1), weight.size(2))
    assert K == rb.K, (K, rb.K)
    assert weight.is_contiguous() and inp.size(1) == Ci
    uniq = int(rb.unique_col is not None and (rb.unique_col == out_col or rb.unique_col == 2))
    # `inp` may be a column slice of a wider row-major tensor (the right half of a joined gradient): the pipelined kernels
    # read it in place through its leading dimension; `out_view`: a preallocated slice the result is written into (the
    # right half of a joined tensor; the one-contribution-per-row kernel only).  The library plans a strided input by the
    # rows its span WOULD hold at C_i planes (scn_conv_forward): the plan, workspace and statistics queries below are asked
    # with the same number, and where that plan cannot read a slice the input is copied (round-3 advisor).
    ld_in = 0
    n_plan = inp.size(0)
    if not inp.is_contiguous():
        if _row_slice(inp):
            n_eff = (inp.size(0) * inp.stride(0) + Ci - 1) // Ci
            if lib.scn_conv_plan(K, rb.stride, n_eff, n_out, Ci, Co, uniq) // 100 in (2, 4, 5):
                ld_in, n_plan = inp.stride(0), n_eff
        if not ld_in:
            inp = inp.contiguous()
    code, ws_bytes, nblk_lib = _plan(lib, K, rb.stride, n_plan, n_out, Ci, Co, uniq)
    kind = code // 100
    ld_out = 0
    if out_view is not None:
        assert kind == 5 and _row_slice(out_view) and tuple(out_view.shape) == (n_out, Co), (kind, out_view.shape)
        out, ld_out = out_view, out_view.stride(0)
    else:
        out.resize_(n_out, Co)
    if residual is not None:
        assert residual.is_contiguous() and tuple(residual.shape) == (n_out, Co) and residual.data_ptr() != out.data_ptr()
    st = stream()  # (once per call: torch.cuda.current_stream() is 2 us of host time, and the forward-only path is bound there)
    wpack, ready = _packed_weight(weight, w_transposed, w_reverse, K, Ci, Co, st)
    # launches with few output tiles split their offsets over workgroups; partial tiles go through this workspace
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=inp.device) if ws_bytes else None
    # column statistics of the result for the batch norm that consumes it (stats_out: a list that receives the tensor)
    cstats = None
    bnb = None
    if bn_link is not None and not (BN_BWD_STATS_FROM_CONV and tuple(bn_link.x.shape) == (n_out, Co)
                                    and bn_link.x.is_contiguous()):
        bn_link = None
    if stats_out is not None or bn_link is not None:
        nblk = nblk_lib
        if kind in (2, 4) and not SPLIT_FUSED and code % 100 > 1:
            nblk = 0  # (a split launch emits statistics only when it adds its partial tiles itself: sk_state)
        if nblk > 0:
            cstats = torch.empty((nblk, 2, Co), dtype=torch.float32, device=inp.device)
            if bn_link is not None:
                bnb = bn_link.struct()
            else:
                stats_out.append(cstats)
    if w_transposed and PREFETCH_AT_COARSE_LEVELS and (kind == 2 or (kind == 4 and code % 100 > 1)):
        # the backward pass has reached the coarse levels: the point a rulebook prefetch waits for (prefetch_input)
        mk = _COARSE_MARK.get(inp.device.index)
        if mk is None:
            mk = _COARSE_MARK[inp.device.index] = [torch.cuda.Event(), -1]
        if mk[1] != _step_count(inp.device):
            mk[0].record(torch.cuda.current_stream(inp.device))
            mk[1] = _step_count(inp.device)
    passes = None
    if kind == 4 and code % 100 == 1 and STREAM_K and not (STREAM_K_FWD_ONLY and w_transposed) and K > 1 and not _prefetch_in_flight():
        passes = rb.passes.get(out_col)
        if passes is None and (n_out + 127) // 128 >= _SK_MIN_TILES:
            _SK_WANTED.add((K, out_col))  # (from the next batch on the tables come with the rulebook)
            passes = _tile_passes(rb, out_col, n_out)
    ev = _prof_begin()
    check(lib.scn_conv_forward(ptr(inp), ptr(out), ptr(weight), int(w_transposed), int(w_reverse),
                               ptr(bias) if bias is not None and bias.numel() else None, ptr(residual), ptr(cstats),
                               ctypes.byref(bnb) if bnb is not None else None, ptr(rb.rules), ptr(rb.bases[out_col]),
                               ptr(passes),
                               ptr(rb.counts), K, rb.stride,
                               in_col, out_col, uniq,
                               inp.size(0), n_out, Ci, Co, ld_in, ld_out, ptr(wpack), ready, ptr(ws), ws_bytes,
                               ptr(_sk_state(inp.device, st)) if (STREAM_K and kind == 4) or (SPLIT_FUSED and kind in (2, 4) and code % 100 > 1) else None, st))
    if bn_link is not None:
        bn_link.offer(out, cstats if bnb is not None else None)
    # algorithmic work of one launch: flops = 2 R Ci Co;
    # bytes = 4 (R Ci + n_out Co) + 8 R + 4 K Ci Co
    if ev is not None:
        R = rb.total_pairs
        name = _CONV_KERNELS.get(code // 100, "conv") + ("_split" if code % 100 > 1 else "")
        if code // 100 == 0 and Ci <= 4 and Co % 16 == 0 and not w_transposed and residual is None:
            name = "conv_stem"  # k_conv_stem: the narrow-input layer as a GEMM on the matrix cores
        _prof_end(ev, name, 2.0 * R * Ci * Co, 4.0 * (R * Ci + n_out * Co) + 8.0 * R + 4.0 * K * Ci * Co,
                  (K, n_out, Ci, Co, int(w_transposed), in_col))
        fused = 4.0 * n_out * Co * (int(residual is not None) + int(bnb is not None))
        if fused:
            PROFILE_FUSED[name] = PROFILE_FUSED.get(name, 0.0) + fused


# ---- dWeight on a second HIP stream.  dInput and dWeight of a layer both consume dOut and nothing in backward consumes
# dWeight, so inside an autograd backward pass the dWeight kernel is launched on a side stream: it starts when the dInput
# kernel (whose workgroups hold all the LDS of a CU) drains -- filling the tail in which the last tiles run on a mostly
# idle chip -- and keeps running beside the batch-norm backward and the elementwise kernels that follow, which are
# HBM-bound.  The main stream waits for the side stream at the end of the backward pass (autograd final callback) and
# before a gradient bucket is all-reduced (dw_join, occuseg_amd/dp.py).  Outside autograd (the boundary functions called
# directly, as the reference's C++ would) everything stays on the caller's stream.
# Only when the CALLER says nothing reads dWeight before the end of the backward pass (`dw_async`): the convolution
# modules do when they accumulate straight into a gradient buffer they own (occuseg_amd.dp.FlatGradBucket) and return no
# gradient to autograd; on the plain autograd path AccumulateGrad reads the returned tensor at once, on the main stream.
DW_SIDE_STREAM = True   # A/B switch (tools/ab_step.py DW_SIDE_STREAM=0,1)
DW_FIRST = False        # A/B switch: dWeight queued (side stream) before the layer's dInput (main stream)
_DW_STATE = {}          # device index -> [stream, last event or None, [(event, dOut kept alive), ...]]
_DW_ALLOW = [0]


@contextlib.contextmanager
def dw_async(allowed):
    """inside: the backward boundary functions may leave their dWeight kernel running on the side stream.
    The side-stream kernel reads dOut while the main stream goes on.  dOut is autograd's tensor, and the SAME tensor may
    be somebody else's gradient too (AddBackward of a residual block hands one tensor to the body and to the shortcut:
    scn.add_feature_planes / AddTable over ConcatTable, sparseconvnet/networkArchitectures.py:222-229): where it meets
    another gradient autograd adds IN PLACE into a tensor it holds the last reference to (torch/csrc/autograd/
    input_buffer.cpp: can_accumulate_inplace), on the main stream, under the side stream's feet.  Every dOut that goes to
    the side stream is therefore kept referenced until its kernel has finished (then autograd adds out of place: same
    traffic), whatever the graph around the convolution looks like."""
    before = _DW_ALLOW[0]
    _DW_ALLOW[0] = 1 if allowed else 0
    try:
        yield
    finally:
        _DW_ALLOW[0] = before


# HIP stream priority of the dWeight side stream: 0 = a torch stream of normal priority; 1 = HIP's LOW priority (torch's own
# streams know high and normal only: created with hipStreamCreateWithPriority and wrapped) -- the dInput / batch-norm kernels
# of the main stream are the critical path of the backward pass, the dWeight kernels only have to be done at its end.
DW_PRIORITY = int(os.environ.get("SCN_DW_PRIORITY", "0"))
_HIP_RT = []


def _low_priority_stream(dev, priority):
    if not _HIP_RT:
        _HIP_RT.append(ctypes.CDLL("libamdhip64.so"))
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = _HIP_RT[0].hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(priority))  # 1 = hipStreamNonBlocking
    if rc != 0 or not h.value:
        raise RuntimeError("occuseg_amd: hipStreamCreateWithPriority(%d) failed (%d)" % (priority, rc))
    return torch.cuda.ExternalStream(h.value, device=dev)


def _dw_state(dev):
    st = _DW_STATE.get(dev.index)
    if st is None:
        side = _low_priority_stream(dev, DW_PRIORITY) if DW_PRIORITY > 0 else torch.cuda.Stream(device=dev)
        st = _DW_STATE[dev.index] = [side, None, []]
    return st


def _dw_side_begin(*tensors):
    if not (DW_SIDE_STREAM and _DW_ALLOW[0]):
        return None
    dev = tensors[0].device
    st = _dw_state(dev)
    try:
        # Only inside a backward pass; this is what makes the main stream wait for the side stream at its end.  Queued
        # at EVERY launch (the callback is idempotent): a flag that says "already queued" stays set for ever when a
        # backward pass raises -- final callbacks are then skipped -- and no later pass would join the side stream.
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _dw_finish(dev.index))
    except RuntimeError:
        return None
    side = st[0]
    side.wait_stream(torch.cuda.current_stream(dev))  # dOut, the zeroed dW, the saved input: produced on the main stream
    for t in tensors:
        if t is not None:
            t.record_stream(side)  # (the caching allocator must not hand their memory out before the side stream is done)
    held = st[2]
    while held and held[0][0].query():  # kernels that have finished no longer pin their dOut
        held.pop(0)
    return side


def _dw_side_end(side, d_out, rb=None):
    """`rb`: the rulebook the kernel walks.  It is held like dOut: the LAST reference to a batch's Metadata_3 -- hence to
    its rulebooks -- is dropped INSIDE the backward pass (InputLayer's backward clears it), before the main stream has
    joined the side stream; with the rulebooks allocated on the prefetch stream (SCN.prefetch_input) their memory went
    back to that stream's pool as soon as the MAIN stream had passed, and the next batch's rule build overwrote the pair
    lists under the last dWeight kernels (round 3: GPU memory access faults in 5 of 30 runs of 42 steps on one box, first
    seen under rocprofv3 only; never with the prefetch or the side stream off)."""
    st = _DW_STATE[side.device.index]
    ev = torch.cuda.Event()
    ev.record(side)
    st[1] = ev
    st[2].append((ev, d_out, rb))


def dw_join(device=None):
    """make the current stream wait for every dWeight kernel launched so far (called before gradients are consumed or
    zeroed: FlatGradBucket.zero / _launch / all_reduce_mean, the end of every backward pass)"""
    for idx, st in _DW_STATE.items():
        if device is None or device.index is None or device.index == idx:
            if st[1] is not None:
                torch.cuda.current_stream(torch.device("cuda", idx)).wait_event(st[1])
            while st[2] and st[2][0][0].query():
                st[2].pop(0)


def dw_wait_on(stream, device=None):
    """make `stream` (not the current one) wait for every dWeight kernel queued so far: the gradient exchange of a bucket is
    issued from a stream of its own (occuseg_amd.dp.FlatGradBucket._launch) so that the main stream does not stand still"""
    for idx, st in _DW_STATE.items():
        if device is None or device.index is None or device.index == idx:
            if st[1] is not None:
                stream.wait_event(st[1])


def _dw_finish(idx):
    st = _DW_STATE[idx]
    if st[1] is not None:
        torch.cuda.current_stream(torch.device("cuda", idx)).wait_event(st[1])
        st[1] = None
    # the main stream now waits for every side-stream kernel: whatever autograd does with the dOut tensors from here on
    # is ordered behind them
    del st[2][:]


DW_SIDE_TEST_DELAY_CYCLES = 0  # tests only: a spin kernel in front of every side-stream dWeight kernel, so that the side stream
                               # lags the main stream by milliseconds and anything the kernel reads that is not pinned is gone
                               # by the time it runs (tests/test_gpu_memory_safety.py)
# Parameter gradients bitwise reproducible run to run (on by default since round 4; SCN_DETERMINISTIC_DW=0 / False: float atomics
# as the reference, sparseconvnet/SCN/CUDA/Convolution.cu:749).  scn_conv_dweight gets a workspace -- a slot per workgroup -- and
# a ticket block: k_dw_tile adds the slots in a fixed order INSIDE the kernel (no memset, no second launch: the first form, memset +
# kernel + reduction kernel, cost 0.8 ms per train step in 74 extra launches; this one 0.05-0.2 ms, profiles/r04_ab_deterministic_dw.txt).
DETERMINISTIC_DW = os.environ.get("SCN_DETERMINISTIC_DW", "1") != "0"


_DW_TICKETS = {}


def _dw_ticket_state(device):
    """SCN_DW_STATE_BYTES of zeros per (device, stream): the arrival tickets of scn_conv_dweight's in-kernel fixed-order
    reduction.  Every launch leaves the block as it found it; launches on one stream never overlap."""
    key = (device.index, stream())
    t = _DW_TICKETS.get(key)
    if t is None:
        t = _DW_TICKETS[key] = torch.zeros(_lib.DW_STATE_BYTES, dtype=torch.uint8, device=device)
    return t


def _dw_workspace(rb, out_col, K, n_in, n_dout, Ci, Co, device):
    if not DETERMINISTIC_DW:
        return None, 0
    nbytes = _lib_().scn_conv_dweight_workspace_bytes(K, rb.stride, n_in, n_dout, Ci, Co, int(rb.bases.get(out_col) is not None))
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def _conv_backward(inp, d_in, d_out, weight, d_weight, d_bias, rb, in_col, out_col, n_in, reverse, bn_link=None,
                   d_in_add=None):
    """d_in = gather-mm-scatter of d_out with the transposed weights; dW; d_bias.
    bn_link: see BnLink; d_in_add: a tensor added to d_in inside the dInput kernel (its `residual`)."""
    lib = _lib_()
    K, Ci, Co = weight.shape
    # d_out may be a column slice of a wider gradient (the right half of what a JoinTable received): read in place by the
    # dInput launch (_conv) and, for planes that are multiples of 32, by the dWeight kernel
    ld_dout = 0
    if not d_out.is_contiguous():
        if _row_slice(d_out) and Ci % 32 == 0 and Co % 32 == 0 and not (d_bias is not None and d_bias.numel()):
            ld_dout = d_out.stride(0)
        else:
            d_out = d_out.contiguous()
    _BWD_SEEN[0] = True  # (weight epochs: the next forward-direction call may see updated weights)

    def d_input():
        if d_in is None:
            return
        if reverse:
            # submanifold: dIn[i] = sum_k' dOut[nbr(i,k')] . W[26-k']^T over the SAME lists (rules are
            # symmetric: (i,o) in list k <=> (o,i) in list 26-k), so the columns keep their roles
            _conv(d_out, d_in, weight, None, rb, in_col, out_col, n_in, True, True, residual=d_in_add, bn_link=bn_link)
        else:
            _conv(d_out, d_in, weight, None, rb, out_col, in_col, n_in, True, False, residual=d_in_add, bn_link=bn_link)

    # DW_FIRST: the side stream's wait on the main stream is taken BEFORE the dInput kernel is queued, so the two kernels
    # of a layer run beside each other from the start (A/B switch; default: dWeight waits for the layer's dInput)
    if not DW_FIRST:
        d_input()
    if d_weight is not None and d_weight.numel():
        # (everything the kernel reads or writes: the rulebook's tensors too -- see _dw_side_end)
        side = _dw_side_begin(inp, d_out, d_weight, rb.rules, rb.counts, rb.bases.get(out_col))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if side is not None and DW_SIDE_TEST_DELAY_CYCLES:
                torch.cuda._sleep(int(DW_SIDE_TEST_DELAY_CYCLES))
            ev = _prof_begin()
            ws, wsb = _dw_workspace(rb, out_col, K, inp.size(0), d_out.size(0), Ci, Co, inp.device)
            check(lib.scn_conv_dweight(ptr(inp), ptr(d_out), ptr(d_weight), ptr(rb.rules), ptr(rb.counts),
                                       ptr(rb.bases.get(out_col)), K, rb.stride, in_col, out_col, inp.size(0),
                                       d_out.size(0), Ci, Co, 0, ld_dout, ptr(ws), wsb, ptr(_dw_ticket_state(inp.device)) if ws is not None else None,
                                       stream()))
            R = rb.total_pairs
            _prof_end(ev, "dw_tile" if Ci % 32 == 0 and Co % 32 == 0 else
                      ("dw_stem" if Ci <= 4 and Co % 16 == 0 and rb.bases.get(out_col) is not None else "dw_narrow"), 2.0 * R * Ci * Co,
                      4.0 * R * (Ci + Co) + 8.0 * R + 4.0 * K * Ci * Co, (K, inp.size(0), Ci, Co, 0, in_col))
        if side is not None:
            _dw_side_end(side, d_out, rb)
    if DW_FIRST:
        d_input()
    if d_bias is not None and d_bias.numel():
        check(lib.scn_column_sum(ptr(d_out), ptr(d_bias), d_out.size(0), Co, stream()))


# ---------------------------------------------------- boundary functions ---
def n_rulebook_bits():
    return 32


def InputLayer_updateOutput(m, spatialSize, input_coords, input_features, output_features, batchSize, mode,
                            input_normal=None):
    """sparseconvnet/SCN/pybind.cpp:158-162 -> sparseconvnet/SCN/CUDA/IOLayers.cpp:17-80"""
    _need_cuda(input_features, "InputLayer_updateOutput")
    # use_normal = a [P,3] tensor with one normal per input row (sparseconvnet/SCN/CUDA/IOLayers.cpp:38-40); anything
    # else (the reference's callers pass zeros(3,3) or an empty tensor) means "no normals"
    use_normal = (input_normal is not None and torch.is_tensor(input_normal) and input_normal.dim() == 2
                  and input_normal.size(0) == input_features.size(0) and input_normal.size(1) == 3)
    m.inputLayer(spatialSize, input_coords, batchSize, mode, input_features.device,
                 normals=input_normal if use_normal else None)
    N = m.nActive[m.inputSpatialSize]
    C = input_features.size(1)
    output_features.resize_(N, C)
    ev = _prof_begin("io")
    check(_lib_().scn_input_forward(ptr(input_features.contiguous()), ptr(output_features), ptr(m.input_order),
                                    ptr(m.input_site_start), N, C, int(mode == 4), stream()))
    _prof_end(ev, "io", 0.0, 4.0 * C * (m.input_P + N) + 4.0 * (m.input_P + N), (m.input_P, N, C))


def InputLayer_updateGradInput(m, d_input_features, d_output_features):
    """sparseconvnet/SCN/CUDA/IOLayers.cpp:81-106"""
    _need_cuda(d_output_features, "InputLayer_updateGradInput")
    C = d_output_features.size(1)
    d_input_features.resize_(m.input_P, C)
    ev = _prof_begin("io")
    check(_lib_().scn_input_backward(ptr(d_input_features), ptr(d_output_features.contiguous()),
                                     ptr(m.input_site_of_row), ptr(m.input_site_start), m.input_P, C,
                                     int(m.input_mode == 4), stream()))
    _prof_end(ev, "io", 0.0, 8.0 * C * m.input_P + 12.0 * m.input_P, (m.input_P, 0, C))


def OutputLayer_updateOutput(m, input_features, output_features):
    """sparseconvnet/SCN/CUDA/IOLayers.cpp:108-129: every original point gets its site's row"""
    _need_cuda(input_features, "OutputLayer_updateOutput")
    C = input_features.size(1)
    output_features.resize_(m.input_P, C)
    ev = _prof_begin("io")
    check(_lib_().scn_output_forward(ptr(input_features.contiguous()), ptr(output_features),
                                     ptr(m.input_site_of_row), m.input_P, C, stream()))
    _prof_end(ev, "io", 0.0, 8.0 * C * m.input_P + 4.0 * m.input_P, (m.input_P, 0, C))


def OutputLayer_updateGradInput(m, d_input_features, d_output_features):
    """sparseconvnet/SCN/CUDA/IOLayers.cpp:131-154"""
    _need_cuda(d_output_features, "OutputLayer_updateGradInput")
    C = d_output_features.size(1)
    N = m.nActive[m.inputSpatialSize]
    d_input_features.resize_(N, C)
    ev = _prof_begin("io")
    check(_lib_().scn_output_backward(ptr(d_input_features), ptr(d_output_features.contiguous()),
                                      ptr(m.input_order), ptr(m.input_site_start), N, C, stream()))
    _prof_end(ev, "io", 0.0, 4.0 * C * (m.input_P + N) + 4.0 * (m.input_P + N), (m.input_P, N, C))


def SubmanifoldConvolution_updateOutput(spatialSize, filterSize, m, input_features, output_features, weight,
                                        bias, dilated_rate=1, residual=None, stats_out=None):
    """sparseconvnet/SCN/pybind.cpp:138-142 -> sparseconvnet/SCN/CUDA/Convolution.cpp:105-162.
    Returns the multiply-add count sum_k R_k*Ci*Co (the reference returns 0 on
    its chunk path, sparseconvnet/SCN/CUDA/Convolution.cu:1286,1374 -- treated as a bug).
    ``residual`` (not in the reference): [N, nOut] tensor added to the result inside the kernel.  ``stats_out`` (not in
    the reference): a list that receives the per-block column sums / sums of squares of the result, which
    ``BatchNormalization_updateOutput(..., _colstats=...)`` consumes instead of reading the tensor once more."""
    _need_cuda(input_features, "SubmanifoldConvolution_updateOutput")
    rb = m.getSubmanifoldRuleBook(spatialSize, filterSize, dilated_rate)
    n = m.getNActive(spatialSize)
    _conv(input_features.contiguous(), output_features, weight, bias, rb, 0, 1, n,
          residual=residual.contiguous() if residual is not None else None, stats_out=stats_out)
    return float(rb.total_pairs) * weight.size(1) * weight.size(2)


def SubmanifoldConvolution_backward(spatialSize, filterSize, m, input_features, d_input_features,
                                    d_output_features, weight, d_weight, d_bias, dilated_rate=1, _bn_link=None):
    """sparseconvnet/SCN/pybind.cpp:143-147 -> sparseconvnet/SCN/CUDA/Convolution.cpp:165-210.
    d_input_features may be None to skip the dInput half."""
    _need_cuda(d_output_features, "SubmanifoldConvolution_backward")
    rb = m.getSubmanifoldRuleBook(spatialSize, filterSize, dilated_rate)
    n = m.getNActive(spatialSize)
    if rb.ori is None:
        _conv_backward(input_features.contiguous(), d_input_features, d_output_features, weight, d_weight, d_bias,
                       rb, 0, 1, n, reverse=True, bn_link=_bn_link)
        return
    # Normal-guided rulebook: list n holds the pairs found at offset i = inv(class of the OUTPUT site)[n], so the
    # (i,o) <-> (o,i) symmetry the plain dInput relies on is gone and an input site can appear several times in one
    # list.  Split by class instead: over the outputs of class c the lists are the plain offsets i (input site unique
    # and ascending per list) with weights W[rot_index[c][i]]; dIn = sum over the three classes.
    _conv_backward(input_features.contiguous(), None, d_output_features, weight, d_weight, d_bias, rb, 0, 1, n,
                   reverse=False)
    if d_input_features is not None:
        d_out = d_output_features.contiguous()
        total = None
        for cls, book in m._class_books(spatialSize, rb).items():
            w_c = weight.index_select(0, torch.from_numpy(ROT_INDEX[cls]).to(weight.device))
            part = torch.empty(0, dtype=d_out.dtype, device=d_out.device)
            _conv(d_out, part, w_c, None, book, 1, 0, n, True, False)
            total = part if total is None else total.add_(part)
        d_input_features.resize_(n, weight.size(1)).copy_(total)


def Convolution_updateOutput(inputSize, outputSize, filterSize, filterStride, m, input_features,
                             output_features, weight, bias, stats_out=None):
    """sparseconvnet/SCN/pybind.cpp:58-63 -> sparseconvnet/SCN/CUDA/Convolution.cpp:36-66"""
    _need_cuda(input_features, "Convolution_updateOutput")
    rb = m.getRuleBook(inputSize, outputSize, filterSize, filterStride)
    n_out = m.getNActive(outputSize)
    _conv(input_features.contiguous(), output_features, weight, bias, rb, 0, 1, n_out, stats_out=stats_out)
    return float(rb.total_pairs) * weight.size(1) * weight.size(2)


def Convolution_backward(inputSize, outputSize, filterSize, filterStride, m, input_features, d_input_features,
                         d_output_features, weight, d_weight, d_bias, _bn_link=None):
    """sparseconvnet/SCN/pybind.cpp:64-69 -> sparseconvnet/SCN/CUDA/Convolution.cpp:69-102"""
    _need_cuda(d_output_features, "Convolution_backward")
    rb = m.getRuleBook(inputSize, outputSize, filterSize, filterStride)
    n_in = m.getNActive(inputSize)
    _conv_backward(input_features.contiguous(), d_input_features, d_output_features, weight, d_weight, d_bias,
                   rb, 0, 1, n_in, reverse=False, bn_link=_bn_link)


def deconvolution_can_write_a_slice(inputSize, outputSize, filterSize, filterStride, m, nIn, nOut):
    """True when Deconvolution_updateOutput(..., _out_view) can write its result into a column slice of a wider tensor"""
    rb = m.getRuleBook(outputSize, inputSize, filterSize, filterStride)
    uniq = int(rb.unique_col is not None and (rb.unique_col == 0 or rb.unique_col == 2))
    return _lib_().scn_conv_plan(rb.K, rb.stride, m.getNActive(inputSize), m.getNActive(outputSize), nIn, nOut, uniq) // 100 == 5


def Deconvolution_updateOutput(inputSize, outputSize, filterSize, filterStride, m, input_features,
                               output_features, weight, bias, stats_out=None, _out_view=None):
    """sparseconvnet/SCN/pybind.cpp:82-87 -> sparseconvnet/SCN/CUDA/Deconvolution.cpp:20-49: the conv rulebook
    of (outputSize -> inputSize) with the columns swapped."""
    _need_cuda(input_features, "Deconvolution_updateOutput")
    rb = m.getRuleBook(outputSize, inputSize, filterSize, filterStride)
    n_out = m.getNActive(outputSize)
    _conv(input_features.contiguous(), output_features, weight, bias, rb, 1, 0, n_out, stats_out=stats_out,
          out_view=_out_view)
    return float(rb.total_pairs) * weight.size(1) * weight.size(2)


def Deconvolution_backward(inputSize, outputSize, filterSize, filterStride, m, input_features,
                           d_input_features, d_output_features, weight, d_weight, d_bias, _bn_link=None):
    """sparseconvnet/SCN/pybind.cpp:88-93 -> sparseconvnet/SCN/CUDA/Deconvolution.cpp:52-84"""
    _need_cuda(d_output_features, "Deconvolution_backward")
    rb = m.getRuleBook(outputSize, inputSize, filterSize, filterStride)
    n_in = m.getNActive(inputSize)
    _conv_backward(input_features.contiguous(), d_input_features, d_output_features, weight, d_weight, d_bias,
                   rb, 1, 0, n_in, reverse=False, bn_link=_bn_link)


def _bn_temp(C, device):
    tb = _lib_().scn_bn_temp_bytes(C)
    return torch.empty(tb, dtype=torch.uint8, device=device)


def BatchNormalization_updateOutput(input_features, output_features, saveMean, saveInvStd, runningMean,
                                    runningVar, weight, bias, eps, momentum, train, leakiness, _colstats=None):
    """sparseconvnet/SCN/sparseconvnet.h:21-26 -> sparseconvnet/SCN/CUDA/BatchNormalization.cpp:21-42.
    ``_colstats`` (not in the reference): [(planes, partial sums [blocks, 2, planes]), ...] covering all planes in order,
    as the producing convolutions emitted them: the statistics pass over the input is skipped."""
    _need_cuda(input_features, "BatchNormalization_updateOutput")
    x = input_features.contiguous()
    N, C = x.shape
    output_features.resize_(N, C)
    saveMean.resize_(C)
    saveInvStd.resize_(C)
    temp = _bn_temp(C, x.device)
    ev = _prof_begin("bn")
    if train and _colstats:
        assert sum(n for n, _ in _colstats) == C and all(p.is_contiguous() and p.size(2) == n for n, p in _colstats)
        nseg = len(_colstats)
        ncols = (ctypes.c_int32 * nseg)(*[n for n, _ in _colstats])
        parts = (ctypes.c_void_p * nseg)(*[p.data_ptr() for _, p in _colstats])
        nblks = (ctypes.c_int32 * nseg)(*[p.size(0) for _, p in _colstats])
        st = stream()
        check(_lib_().scn_bn_forward_ext(ptr(x), ptr(output_features), ptr(saveMean), ptr(saveInvStd), ptr(runningMean),
                                         ptr(runningVar), ptr(weight), ptr(bias), float(eps), float(momentum),
                                         float(leakiness), N, C, ptr(temp), nseg, ncols, parts, nblks,
                                         ptr(_sk_state(x.device, st)) if BN_FOLD_FUSED else None, st))
        _prof_end(ev, "bn_fwd", 0.0, 8.0 * N * C, (N, C, "ext"))  # x read once, y written
        return
    check(_lib_().scn_bn_forward(ptr(x), ptr(output_features), ptr(saveMean), ptr(saveInvStd), ptr(runningMean),
                                 ptr(runningVar), ptr(weight), ptr(bias), float(eps), float(momentum), int(train),
                                 float(leakiness), N, C, ptr(temp), stream()))
    _prof_end(ev, "bn_fwd", 0.0, (12.0 if train else 8.0) * N * C, (N, C))  # read, read, write


def BatchNormalization_backward(input_features, d_input_features, output_features, d_output_features, saveMean,
                                saveInvStd, runningMean, runningVar, weight, bias, d_weight, d_bias, leakiness,
                                _rewrite_d_output=True, _accumulate=False, _mask_from_x=False, _partials=None, _d_x_add=None):
    """sparseconvnet/SCN/sparseconvnet.h:27-32 -> sparseconvnet/SCN/CUDA/BatchNormalization.cpp:45-76.
    d_output_features must be contiguous: it is rewritten in place
    (sparseconvnet/SCN/CUDA/BatchNormalization.cu:151-153).  ``_rewrite_d_output=False`` (not in the reference;
    used by the autograd module, which must not mutate autograd's buffer) leaves it untouched; ``_accumulate=True``
    adds to d_weight / d_bias instead of assigning; ``_mask_from_x=True`` recomputes the sign of y from x instead of
    reading ``output_features`` (bit-identical, 20 instead of 28 bytes per element; implies d_output untouched)."""
    _need_cuda(d_output_features, "BatchNormalization_backward")
    assert d_output_features.is_contiguous()
    x = input_features.contiguous()
    N, C = x.shape
    d_input_features.resize_(N, C)
    temp = _bn_temp(C, x.device)
    ev = _prof_begin("bn")
    if _partials is not None or _d_x_add is not None:
        # (not in the reference) statistics handed over by the kernel that produced d_output_features, and / or another
        # gradient of input_features added to the result inside the apply pass
        assert _mask_from_x
        assert _partials is None or (_partials.is_contiguous() and tuple(_partials.shape[1:]) == (2, C))
        assert _d_x_add is None or (tuple(_d_x_add.shape) == (N, C) and (_d_x_add.is_contiguous() or _row_slice(_d_x_add)))
        st = stream()
        check(_lib_().scn_bn_backward_ext(ptr(x), ptr(d_input_features), ptr(d_output_features), ptr(saveMean),
                                          ptr(saveInvStd), ptr(weight), ptr(bias), ptr(d_weight), ptr(d_bias),
                                          float(leakiness), N, C, 2 if _accumulate else 0, ptr(temp), ptr(_partials),
                                          _partials.size(0) if _partials is not None else 0, ptr(_d_x_add),
                                          _d_x_add.stride(0) if _d_x_add is not None else 0,
                                          ptr(_sk_state(x.device, st)) if BN_FOLD_FUSED and _partials is not None else None, st))
        _prof_end(ev, "bn_bwd", 0.0, (12.0 if _partials is not None else 20.0) * N * C + (4.0 * N * C if _d_x_add is not None else 0.0),
                  (N, C, "ext"))
        return
    flags = (4 if _mask_from_x else int(bool(_rewrite_d_output))) | (2 if _accumulate else 0)
    check(_lib_().scn_bn_backward(ptr(x), ptr(d_input_features), ptr(output_features), ptr(d_output_features),
                                  ptr(saveMean), ptr(saveInvStd), ptr(weight), ptr(bias), ptr(d_weight), ptr(d_bias),
                                  float(leakiness), N, C, flags, ptr(temp), stream()))
    _prof_end(ev, "bn_bwd", 0.0, 20.0 * N * C, (N, C))  # x, y, dOut read; dOut, dIn written


_IDENTITY_RB = {}  # (n, device index) -> _RuleBook with the single list {(i, i)}; tiny LRU
NIN_USE_GATHER_MM = True


def _identity_rulebook(n, device):
    """1x1 convolution == the gather-mm kernels over the identity rulebook (what list 13 of every
    submanifold rulebook is).  Lets NetworkInNetwork share the fp32-MFMA kernels; rocBLAS' x^T.dy
    with a 600k-long reduction ran at ~19 TFLOP/s, the dWeight kernel here runs it at ~100."""
    key = (n, device.index)
    rb = _IDENTITY_RB.pop(key, None)
    if rb is None:
        idx = torch.arange(n, dtype=torch.int32, device=device)
        rules = torch.stack([idx, idx], 1).reshape(1, n, 2).contiguous()
        rb = _RuleBook(1, n, rules, torch.tensor([n], dtype=torch.int32, device=device))
        n_gran = (n + GRANULE - 1) // GRANULE
        bases = torch.clamp(torch.arange(n_gran + 1, dtype=torch.int32, device=device) * GRANULE, max=n)
        rb.bases[0] = rb.bases[1] = bases.reshape(1, n_gran + 1).contiguous()
        rb.total_pairs = n
        rb.unique_col = 2
        while len(_IDENTITY_RB) >= 64:  # (two batches x five levels of the m64 UNet already are ten sizes)
            _IDENTITY_RB.pop(next(iter(_IDENTITY_RB)))
    _IDENTITY_RB[key] = rb
    return rb


def _nin_native(x, weight):
    return (NIN_USE_GATHER_MM and x.is_cuda and x.dtype == torch.float32 and x.size(0) > 0
            and weight.size(0) % 32 == 0 and weight.size(1) % 32 == 0)


def NetworkInNetwork_updateOutput(input_features, output_features, weight, bias):
    """sparseconvnet/SCN/CUDA/NetworkInNetwork.cpp:9-25 (addmm).  Planes that are multiples of 32 go through the
    gather-mm kernel over the identity rulebook, anything else through a library GEMM (rocBLAS via torch),
    as the reference calls cuBLAS through ATen."""
    n = input_features.size(0)
    if _nin_native(input_features, weight):
        rb = _identity_rulebook(n, input_features.device)
        _conv(input_features.contiguous(), output_features, weight.view(1, weight.size(0), weight.size(1)), bias, rb,
              0, 1, n)
        return float(n) * weight.size(0) * weight.size(1)
    output_features.resize_(n, weight.size(1))
    if bias is not None and bias.numel():
        output_features.copy_(bias.expand_as(output_features))
        if n:
            output_features.addmm_(input_features, weight)
    elif n:
        torch.mm(input_features, weight, out=output_features)
    return float(n) * weight.size(0) * weight.size(1)


def NetworkInNetwork_updateGradInput(d_input_features, d_output_features, weight):
    """sparseconvnet/SCN/CUDA/NetworkInNetwork.cpp:27-37"""
    n = d_output_features.size(0)
    _BWD_SEEN[0] = True
    if _nin_native(d_output_features, weight):
        rb = _identity_rulebook(n, d_output_features.device)
        _conv(d_output_features.contiguous(), d_input_features, weight.view(1, weight.size(0), weight.size(1)), None,
              rb, 0, 1, n, w_transposed=True)
        return
    d_input_features.resize_(n, weight.size(0))
    if n:
        torch.mm(d_output_features, weight.t(), out=d_input_features)


def NetworkInNetwork_backward(input_features, d_input_features, d_output_features, weight, d_weight, d_bias,
                              _d_in_add=None):
    """(not in the reference: its three functions below, fused for the autograd module)  dInput (+ ``_d_in_add`` inside
    the kernel), dWeight ACCUMULATED into d_weight (zeroed by the caller, or the owner's flat gradient view: the dWeight
    kernel may then run on the side stream like a convolution's), dBias assigned."""
    n = input_features.size(0)
    if _nin_native(input_features, weight):
        rb = _identity_rulebook(n, input_features.device)
        Ci, Co = weight.shape
        _conv_backward(input_features.contiguous(), d_input_features, d_output_features, weight.view(1, Ci, Co),
                       d_weight.view(1, Ci, Co), d_bias, rb, 0, 1, n, reverse=False, d_in_add=_d_in_add)
        return
    _BWD_SEEN[0] = True
    d_out = d_output_features.contiguous()
    if d_input_features is not None:
        d_input_features.resize_(n, weight.size(0))
        if n:
            torch.mm(d_out, weight.t(), out=d_input_features)
            if _d_in_add is not None:
                d_input_features.add_(_d_in_add)
    if n:
        d_weight.addmm_(input_features.t(), d_out)
        if d_bias is not None and d_bias.numel():
            torch.sum(d_out, 0, out=d_bias)


def NetworkInNetwork_accGradParameters(input_features, d_output_features, d_weight, d_bias):
    """sparseconvnet/SCN/CUDA/NetworkInNetwork.cpp:39-50"""
    n = input_features.size(0)
    _BWD_SEEN[0] = True
    if n and d_bias is not None and d_bias.numel():
        torch.sum(d_output_features, 0, out=d_bias)
    if _nin_native(input_features, d_weight):
        rb = _identity_rulebook(n, input_features.device)
        Ci, Co = d_weight.shape
        d_weight.zero_()
        ev = _prof_begin()
        ws, wsb = _dw_workspace(rb, 1, 1, n, n, Ci, Co, input_features.device)
        check(_lib_().scn_conv_dweight(ptr(input_features.contiguous()), ptr(d_output_features.contiguous()),
                                       ptr(d_weight), ptr(rb.rules), ptr(rb.counts), ptr(rb.bases.get(1)), 1, rb.stride,
                                       0, 1, n, n, Ci, Co, 0, 0, ptr(ws), wsb,
                                       ptr(_dw_ticket_state(input_features.device)) if ws is not None else None, stream()))
        _prof_end(ev, "dw_tile", 2.0 * n * Ci * Co, 4.0 * n * (Ci + Co) + 8.0 * n + 4.0 * Ci * Co, (1, n, Ci, Co, 0, 0))
        return
    if n:
        torch.mm(input_features.t(), d_output_features, out=d_weight)


def _not_in_scope(name):
    def f(*a, **k):
        raise NotImplementedError("sparseconvnet.SCN.%s is outside the hot path this backend provides "
                                  "(SURVEY.md section 8)" % name)
    f.__name__ = name
    return f


def ResolutionBasedScattering(m, points_lr, points_hr, stride):
    """sparseconvnet/SCN/pybind.cpp:33-36 -> ResolutionBasedScatteringCuda (sparseconvnet/SCN/Metadata/ConvolutionRules.h:
    327-342): for every high-resolution point the site id (rank of its key) of the low-resolution point at
    points_hr / stride, or -1.  points_*: int tensors [n,3] of ONE sample (as sparseconvnet/utils.py:85-90 calls it);
    returns an int32 tensor [n_hr] on the device."""
    dev = points_hr.device if points_hr.is_cuda else (points_lr.device if points_lr.is_cuda else None)
    if dev is None:
        raise RuntimeError("occuseg_amd.SCN.ResolutionBasedScattering: points must live on the GPU -- no CPU path")
    lr = points_lr.to(dev, torch.int32).contiguous()
    hr = points_hr.to(dev, torch.int32).contiguous()
    assert lr.dim() == 2 and hr.dim() == 2 and lr.size(1) == 3 and hr.size(1) == 3
    out = torch.empty(hr.size(0), dtype=torch.int32, device=dev)
    if hr.size(0) == 0:
        return out
    if lr.size(0) == 0:
        return out.fill_(-1)
    grid, _order, _site_of_row, _start = m._build_grid(lr, 0, 3, 0, lr.size(0))  # the reference's pHash->insert
    check(_lib_().scn_lookup_sites(ptr(grid.keys), grid.n, ptr(hr), 3, hr.size(0), int(stride), 0, ptr(out), stream()))
    return out


for _n in ("ActivePooling_updateOutput", "ActivePooling_updateGradInput",
           "AveragePooling_updateOutput", "AveragePooling_updateGradInput", "MaxPooling_updateOutput",
           "MaxPooling_updateGradInput", "SparseToDense_updateOutput", "SparseToDense_updateGradInput",
           "FullConvolution_updateOutput", "FullConvolution_backward", "RandomizedStrideConvolution_updateOutput",
           "RandomizedStrideConvolution_backward", "RandomizedStrideMaxPooling_updateOutput",
           "RandomizedStrideMaxPooling_updateGradInput", "PermutohedralSubmanifoldConvolution_updateOutput",
           "PermutohedralSubmanifoldConvolution_backward", "BLInputLayer_updateOutput",
           "BLInputLayer_updateGradInput", "BLOutputLayer_updateOutput", "BLOutputLayer_updateGradInput",
           "UnPooling_updateOutput", "UnPooling_updateGradInput", "AffineReluTrivialConvolution_updateOutput",
           "AffineReluTrivialConvolution_backward", "BatchwiseMultiplicativeDropout_updateOutput",
           "BatchwiseMultiplicativeDropout_updateGradInput", "LeakyReLU_updateOutput", "LeakyReLU_updateGradInput"):
    globals()[_n] = _not_in_scope(_n)


# ---- tracing: a roctx range around every boundary function, as the reference's EASY_FUNCTION / EASY_BLOCK ranges
# (sparseconvnet/SCN/CUDA/Convolution.cpp:112-154, easy_profiler) -- so that `rocprofv3 --kernel-trace --marker-trace` attributes
# the kernels of a step to layers.  Off unless SCN_ROCTX=1 (or SCN.enable_roctx()): then every call costs two ctypes calls.
_ROCTX = [None]
_TRACED = ("InputLayer_updateOutput", "InputLayer_updateGradInput", "OutputLayer_updateOutput", "OutputLayer_updateGradInput",
           "SubmanifoldConvolution_updateOutput", "SubmanifoldConvolution_backward", "Convolution_updateOutput",
           "Convolution_backward", "Deconvolution_updateOutput", "Deconvolution_backward", "BatchNormalization_updateOutput",
           "BatchNormalization_backward", "NetworkInNetwork_updateOutput", "NetworkInNetwork_updateGradInput",
           "NetworkInNetwork_backward")


def _traced(name, fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        lib = _ROCTX[0]
        if lib is None:
            return fn(*args, **kwargs)
        # the range's label: the function and the shapes of its feature / weight tensors, e.g.
        # "SubmanifoldConvolution_updateOutput [601674x64] [27x64x64]"
        shapes = " ".join("[" + "x".join(str(d) for d in a.shape) + "]" for a in args
                          if torch.is_tensor(a) and a.dim() >= 2 and a.numel() and a.is_floating_point())
        lib.roctxRangePushA((name + " " + shapes).encode())
        try:
            return fn(*args, **kwargs)
        finally:
            lib.roctxRangePop()
    wrapper.untraced = fn
    return wrapper


_UNTRACED = {}


def _install_traced(on):
    """swap the boundary functions of this module for their traced wrappers (callers look them up as SCN.<name> at call time);
    off: the plain functions again -- tracing that is off costs nothing, not even a wrapper call (the forward-only configuration
    is bound by the host's launch rate)"""
    g = globals()
    for n in _TRACED:
        if on and n not in _UNTRACED:
            _UNTRACED[n] = g[n]
            g[n] = _traced(n, g[n])
        elif not on and n in _UNTRACED:
            g[n] = _UNTRACED.pop(n)


def enable_roctx(on=True):
    """roctx ranges around the boundary functions on / off; returns whether they are on (False: libroctx64.so not found)"""
    if not on:
        _ROCTX[0] = None
        _install_traced(False)
        return False
    if _ROCTX[0] is None:
        for cand in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                lib = ctypes.CDLL(cand)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _ROCTX[0] = lib
                break
            except (OSError, AttributeError):
                continue
    _install_traced(_ROCTX[0] is not None)
    return _ROCTX[0] is not None


if os.environ.get("SCN_ROCTX", "0") == "1":
    enable_roctx()
