"""The step BEFORE the hot path: the reference's on-disk scenes, its collate functions and a
loader wrapper that hands batches to the GPU backend with their rulebooks already being built.

* ``load_scene(path)``: a ``*_instance.pth`` file as examples/ScanNet/prepare_data.py:95-112 writes it -- a dict
  ``{coords float[P,3], colors float[P,3] in [-1,1], w [P,2] (semantic, instance)[, normals, region, depth]}``.
* ``ScanNetCollate``: the arithmetic of ``ScanNet.trainMerge`` / ``valMerge`` (examples/ScanNet/datasets/scannet.py:
  69-265, 267-424): flip / rotation (/ rotation noise / elastic distortion), scaling to voxels, the +10 voxel origin
  shift, cropping to the ``full_scale`` box, per-instance offsets and displacements, and the batch dict with the
  reference's keys -- in particular ``'x': [locs float[P,4], feats, normals | zeros(3,3), batch]``.  ``locs`` stay
  FLOAT: the backend voxelises them on the GPU (``scn_grid_build(coords_dtype = 2)`` truncates toward zero exactly like
  the reference's ``.type(torch.LongTensor)``, sparseconvnet/ioLayers.py:56), so no int64 copy is made on the host.
* ``PrefetchingLoader``: wraps any iterable of such batches; each batch is moved to the device as it is fetched and
  ``InputLayer.prefetch`` starts building its voxel hash and rulebooks on the side stream at once -- i.e. while the
  PREVIOUS step's kernels are still draining -- so the training loop needs no extra call.
"""
import math

import numpy as np
import torch

REQUIRED_KEYS = ("coords", "colors", "w")


def load_scene(path):
    """torch.load of one ``*_instance.pth`` (examples/ScanNet/datasets/scannet.py:427-433), checked"""
    d = torch.load(path, weights_only=False)
    for k in REQUIRED_KEYS:
        if k not in d:
            raise ValueError("%s: not an *_instance.pth scene (missing '%s')" % (path, k))
    P = len(d["coords"])
    if d["coords"].shape != (P, 3) or d["colors"].shape != (P, 3) or d["w"].shape != (P, 2):
        raise ValueError("%s: expected coords [P,3], colors [P,3], w [P,2]" % path)
    return d


def _instance_ids(w1):
    """instance labels renumbered 0..n-1 in sorted order (np.unique(..., False, True)[1], scannet.py:171)"""
    return np.unique(w1, return_inverse=True)[1].astype(np.int64)


class LegacyRNG(object):
    """The reference's random sources -- the GLOBAL numpy RandomState (np.random.randn / randint / rand / uniform) and
    torch.randn for the colour jitter, seeded with 100 in ScanNet.__init__ (scannet.py:42-45) -- behind the Generator
    method names ScanNetCollate calls.  With it (and the same seeds) the collate functions consume exactly the random
    numbers the reference's trainMerge / valMerge consume, in the same order: tests/golden/g7 compares the two."""

    @staticmethod
    def seed(n=100):
        torch.manual_seed(n)
        np.random.seed(n)

    def standard_normal(self, size=None):
        return np.random.standard_normal(size)

    def integers(self, lo, hi):
        return np.random.randint(lo, hi)

    def random(self, size=None):
        return np.random.random_sample(size)

    def uniform(self, lo, hi, size=None):
        return np.random.uniform(lo, hi, size)

    def colour_noise(self):
        return torch.randn(3).float()


class ScanNetCollate(object):
    """config keys as the reference's (examples/ScanNet/config.py): scale, full_scale, use_normal, use_elastic,
    use_feature ('c' colour, 'l' centred unit vector, 'n' normal, 'd' depth, 'h' height), use_rotation_noise,
    regress_sigma.  ``rng``: numpy Generator (the reference uses the global numpy / torch state seeded with 100)."""

    def __init__(self, config, rng=None):
        self.scale = config["scale"]
        self.full_scale = config["full_scale"]
        self.use_normal = config.get("use_normal", False)
        self.reference_normal_key = config.get("reference_normal_key", True)  # see _scene
        self.use_elastic = config.get("use_elastic", False)
        self.use_feature = config.get("use_feature", "c")
        self.use_rotation_noise = config.get("use_rotation_noise", False)
        self.regress_sigma = config.get("regress_sigma", 0.2)
        self.rng = rng if rng is not None else np.random.default_rng(100)

    # -- augmentation pieces ---------------------------------------------------
    def elastic(self, x, gran, mag):
        """scannet.py:50-67: three smoothed noise fields on a coarse lattice, interpolated at the points"""
        if not self.use_elastic:
            return x
        import scipy.interpolate
        import scipy.ndimage
        blur = [np.ones(s, np.float32) / 3 for s in ((3, 1, 1), (1, 3, 1), (1, 1, 3))]
        bb = np.abs(x).max(0).astype(np.int32) // gran + 3
        noise = [self.rng.standard_normal((bb[0], bb[1], bb[2])).astype(np.float32) for _ in range(3)]
        for _ in range(2):
            for b in blur:
                noise = [scipy.ndimage.convolve(n, b, mode="constant", cval=0) for n in noise]
        ax = [np.linspace(-(b - 1) * gran, (b - 1) * gran, b) for b in bb]
        interp = [scipy.interpolate.RegularGridInterpolator(ax, n, bounds_error=0, fill_value=0) for n in noise]
        return x + np.hstack([i(x)[:, None] for i in interp]) * mag

    def _transform(self, train):
        m = np.eye(3)
        if train and self.use_rotation_noise:
            m = m + self.rng.standard_normal((3, 3)) * 0.1
        m[0][0] *= self.rng.integers(0, 2) * 2 - 1
        m *= self.scale
        theta = self.rng.random() * 2 * math.pi
        return np.matmul(m, [[math.cos(theta), math.sin(theta), 0], [-math.sin(theta), math.cos(theta), 0], [0, 0, 1]])

    # -- one scene ---------------------------------------------------------------
    def _scene(self, s, train):
        a, b, c = np.asarray(s["coords"], np.float64), np.asarray(s["colors"]), np.asarray(s["w"])
        # scannet.py:90-93 tests the key 'normal' but reads 'normals': on standard *_instance.pth files (key 'normals')
        # the reference therefore ALWAYS takes the coordinates as d.  reference_normal_key=True (default) reproduces
        # that; False reads the normals whenever the file has them (the evident intent).
        has_normals = ("normal" in s and "normals" in s) if self.reference_normal_key else ("normals" in s)
        d = np.asarray(s["normals"], np.float64) if has_normals else a
        e = np.asarray(s["depth"]) if "depth" in s else np.asarray(s["coords"])
        region_parts = np.asarray(s["region"]) if "region" in s else c[:, 1]
        m = self._transform(train)
        d = np.matmul(d, m) / self.scale
        if train:
            a = np.matmul(a, m)
            a = self.elastic(a, 6 * self.scale // 50, self.rng.random() * 40 * self.scale / 50)
            a = self.elastic(a, 20 * self.scale // 50, self.rng.random() * 160 * self.scale / 50)
        else:
            a = np.matmul(a, m) + self.full_scale / 2 + self.rng.uniform(-2, 2, 3)
        offset = a.min(0) - 10 + self.rng.random(3)  # the smallest coordinate lands in (9, 10] (scannet.py:133)
        a = a - offset
        idxs = (a.min(1) >= 0) & (a.max(1) < self.full_scale)
        a, b, c, d, e = a[idxs], b[idxs], c[idxs].astype(np.int32), d[idxs], e[idxs]
        region = region_parts[idxs]
        region_index, region_mask = np.unique(region, return_inverse=True)
        c[:, 1] = _instance_ids(c[:, 1])
        inst = c[:, 1].astype(np.int64)
        n_inst = int(inst.max()) + 1 if len(inst) else 0
        size = np.bincount(inst, minlength=n_inst).astype(np.float32)
        a32 = torch.from_numpy(a).float()
        # per-instance centre, offset = exp(-(dist / scale / sigma)^2), displacement = (x - centre) / scale  (:190-200)
        it = torch.from_numpy(inst)
        centre = torch.zeros(n_inst, 3).index_add_(0, it, a32) / torch.from_numpy(size).clamp(min=1).view(-1, 1)
        rel = a32 - centre[it]
        off = torch.exp(-(torch.norm(rel, dim=1) / self.scale / self.regress_sigma) ** 2)
        mask = torch.zeros((len(a), n_inst), dtype=torch.float32)
        mask[torch.arange(len(a)), it] = 1
        lf = a32 - a32.mean(0, keepdim=True)
        feats = []
        if "l" in self.use_feature:
            feats.append(lf / torch.norm(lf, p=2, dim=1, keepdim=True))
        # (the reference draws the colour jitter whether or not colour is a feature, scannet.py:211)
        jitter = self.rng.colour_noise() if hasattr(self.rng, "colour_noise") else \
            torch.from_numpy(self.rng.standard_normal(3)).float()
        if "c" in self.use_feature:
            feats.append(torch.clamp(torch.from_numpy(b).float() + jitter * 0.1, -1, 1))
        if "n" in self.use_feature:
            feats.append(torch.from_numpy(d).float())
        if "d" in self.use_feature:
            feats.append(torch.from_numpy(e).float())
        if "h" in self.use_feature:
            feats.append(a32[:, 2:3])
        return {"a": a32, "feats": torch.cat(feats, 1), "labels": torch.from_numpy(c), "normals": torch.from_numpy(d).float(),
                "idxs": torch.from_numpy(idxs.astype(int)), "mask": mask, "inst": it, "inst_size": torch.from_numpy(size)[it],
                "n_inst": n_inst, "offset": off, "displacement": rel / self.scale,
                "region": torch.from_numpy(region), "region_mask": torch.from_numpy(region_mask),
                "region_index": torch.from_numpy(region_index)}

    # -- the collate functions -----------------------------------------------------
    def merge(self, tbl, scenes, train=True, paths=None, val_offsets=None):
        per = [self._scene(scenes[i], train) for i in tbl]
        locs = torch.cat([torch.cat([p["a"], torch.full((len(p["a"]), 1), float(k))], 1) for k, p in enumerate(per)], 0)
        normals = torch.cat([p["normals"] for p in per], 0) if self.use_normal else torch.zeros(3, 3)
        out = {"x": [locs, torch.cat([p["feats"] for p in per], 0), normals, len(per)],
               "y": torch.cat([p["labels"] for p in per], 0).long(), "id": list(tbl),
               "pth_file": [paths[i] for i in tbl] if paths is not None else [None] * len(per),
               "idxs": [p["idxs"] for p in per], "masks": [p["mask"] for p in per],
               "instance_masks": torch.cat([p["inst"].float() for p in per], 0),
               "instance_sizes": torch.log(torch.cat([p["inst_size"] for p in per], 0)),
               "sizes": torch.stack([torch.tensor(p["n_inst"]) for p in per], 0),
               "offsets": torch.cat([p["offset"] for p in per], 0).view(-1, 1),
               "displacements": torch.cat([p["displacement"] for p in per], 0),
               "regions": torch.cat([p["region"] for p in per], 0),
               "region_masks": torch.cat([p["region_mask"] for p in per], 0),
               "region_indexs": torch.cat([p["region_index"] for p in per], 0)}
        if train:
            out["elastic_locs"] = per[-1]["a"]
        else:
            off = val_offsets if val_offsets is not None else [0] * (max(tbl) + 2)
            out["point_ids"] = torch.cat([torch.nonzero(p["idxs"]).view(-1) + off[i] for i, p in zip(tbl, per)], 0)
        return out

    def trainMerge(self, tbl, train, paths=None):
        return self.merge(tbl, train, True, paths)

    def valMerge(self, tbl, val, valOffsets, paths=None):
        return self.merge(tbl, val, False, paths, valOffsets)


class PrefetchingLoader(object):
    """for batch in PrefetchingLoader(loader, input_layer, device): ... net(batch['x']) ...

    Every batch's ``x = [locs, feats, normals, batch]`` arrives on ``device`` (pinned-memory, non-blocking copies) and
    with its grids + rulebooks already queued on the side stream (``InputLayer.prefetch``), started the moment the batch
    is fetched -- which in a training loop is right after the previous step's kernels were queued, so the build and its
    data-dependent host synchronisations overlap that step.  ``move`` lists further keys of the batch dict to move."""

    def __init__(self, loader, input_layer, device, move=()):
        self.loader, self.input_layer, self.device, self.move = loader, input_layer, torch.device(device), tuple(move)
        self._copy_stream = None
        # record_exposed: per batch, two timed events on the main stream around its wait for the copy stream -- their distance is
        # the time the step stood still for the host -> device copies (bench.py's `h2d.exposed_ms`)
        self.record_exposed, self.exposed = False, []

    def _to_device(self, t):
        if torch.is_tensor(t) and not t.is_cuda:
            t = t.pin_memory() if t.numel() else t
            return t.to(self.device, non_blocking=True)
        return t

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        for batch in self.loader:
            locs, feats, normals, bs = batch["x"]
            use_normal = torch.is_tensor(normals) and normals.dim() == 2 and normals.size(0) == feats.size(0)
            batch = dict(batch)
            main = torch.cuda.current_stream(self.device)
            # host -> device copies on their own stream (on the main stream they would queue behind the previous step),
            # and the build on the side stream waits for exactly these copies -- not for the previous step
            with torch.cuda.stream(self._copy_stream):
                x = [self._to_device(locs.float() if locs.dtype == torch.float64 else locs), self._to_device(feats),
                     self._to_device(normals) if use_normal else normals, bs]
                for k in self.move:
                    batch[k] = self._to_device(batch[k])
                self.input_layer.prefetch(x, wait_for_current_stream=True)
            if self.record_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                main.wait_stream(self._copy_stream)
                e1.record(main)
                self.exposed.append((e0, e1))
            else:
                main.wait_stream(self._copy_stream)
            for t in x[:3] + [batch[k] for k in self.move]:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(main)
            batch["x"] = x
            yield batch
