"""not gpu: SURVEY.md section 8 f2 -- the reference's on-disk scene format and its trainMerge / valMerge collate
(examples/ScanNet/datasets/scannet.py:69-265, 267-424; files written by examples/ScanNet/prepare_data.py:95-112),
checked against the facts the reference code fixes and against its own per-instance loop."""
import numpy as np
import pytest
import torch

REFERENCE_TRAIN_KEYS = {"x", "y", "id", "elastic_locs", "pth_file", "idxs", "masks", "instance_masks", "instance_sizes",
                        "sizes", "offsets", "displacements", "regions", "region_masks", "region_indexs"}


def _scene_file(tmp_path, seed, P=3000, with_normals=False):
    rng = np.random.default_rng(seed)
    d = {"coords": (rng.random((P, 3)) * [4.0, 3.0, 2.5]).astype(np.float32),      # metres
         "colors": rng.uniform(-1, 1, (P, 3)).astype(np.float32),
         "w": np.stack([rng.integers(0, 20, P), rng.choice([3, 7, 8, 100, 250], P)], 1).astype(np.float64)}
    if with_normals:
        n = rng.normal(size=(P, 3))
        d["normals"] = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    path = str(tmp_path / ("scene%04d_00_vh_clean_2_instance.pth" % seed))
    torch.save(d, path)
    return path


def test_scene_file_round_trip_and_validation(tmp_path):
    from occuseg_amd import data
    p = _scene_file(tmp_path, 1)
    s = data.load_scene(p)
    assert set(s) == {"coords", "colors", "w"} and s["coords"].shape == (3000, 3)
    bad = str(tmp_path / "bad_instance.pth")
    torch.save({"coords": np.zeros((4, 3), np.float32)}, bad)
    with pytest.raises(ValueError, match="missing 'colors'"):
        data.load_scene(bad)


@pytest.mark.parametrize("train", [True, False])
def test_merge_matches_the_reference_semantics(tmp_path, train):
    from occuseg_amd import data
    cfg = {"scale": 50, "full_scale": 4096, "use_normal": False, "use_elastic": train, "use_feature": "c",
           "use_rotation_noise": train, "regress_sigma": 0.2}
    paths = [_scene_file(tmp_path, s) for s in (1, 2, 3)]
    scenes = [data.load_scene(p) for p in paths]
    col = data.ScanNetCollate(cfg, rng=np.random.default_rng(7))
    b = col.trainMerge([2, 0], scenes, paths) if train else col.valMerge([2, 0], scenes, [0, 3000, 6000, 9000], paths)
    if train:
        assert set(b) == REFERENCE_TRAIN_KEYS
    else:
        assert set(b) == (REFERENCE_TRAIN_KEYS - {"elastic_locs"}) | {"point_ids"}
    locs, feats, normals, bs = b["x"]
    assert bs == 2 and locs.dtype == torch.float32 and locs.shape[1] == 4 and feats.shape == (len(locs), 3)
    assert torch.equal(normals, torch.zeros(3, 3))                      # use_normal off (scannet.py:250-251)
    assert b["pth_file"] == [paths[2], paths[0]] and b["id"] == [2, 0]
    batch_col = locs[:, 3]
    assert torch.all(batch_col[1:] >= batch_col[:-1]) and set(batch_col.tolist()) == {0.0, 1.0}
    for k in (0, 1):
        a = locs[batch_col == k, :3]
        assert a.min() >= 0 and a.max() < 4096
        assert torch.all(a.min(0)[0] > 9 - 1e-3) and torch.all(a.min(0)[0] <= 10 + 1e-3)  # a.min - 10 + rand(3) (:133)
        # the cloud spans metres * scale voxels
        assert 100 < float(a.max(0)[0][0] - a.min(0)[0][0]) < 400
    # instance ids renumbered 0..n-1 per scene; sizes = distinct instances; instance_sizes = log(point count)
    inst = b["instance_masks"].long()
    off = 0
    for k in (0, 1):
        n = int((batch_col == k).sum())
        ik = inst[off:off + n]
        assert sorted(set(ik.tolist())) == list(range(int(b["sizes"][k])))
        cnt = torch.bincount(ik).float()
        assert torch.allclose(b["instance_sizes"][off:off + n], torch.log(cnt[ik]))
        assert b["masks"][k].shape == (n, int(b["sizes"][k])) and torch.equal(b["masks"][k].argmax(1), ik)
        # the reference's own per-instance loop (scannet.py:190-200) for offsets / displacements
        a = locs[off:off + n, :3]
        for mid in range(int(b["sizes"][k])):
            sel = ik == mid
            mean = a[sel].mean(0)
            want_off = torch.exp(-(torch.norm(a[sel] - mean, dim=1) / 50 / 0.2) ** 2)
            assert torch.allclose(b["offsets"][off:off + n][sel].view(-1), want_off, atol=1e-5)
            assert torch.allclose(b["displacements"][off:off + n][sel], (a[sel] - mean) / 50, atol=1e-4)
        off += n
    assert b["y"].dtype == torch.int64 and b["y"].shape == (len(locs), 2) and int(b["y"][:, 0].max()) < 20
    assert torch.all(feats.abs() <= 1)
    if not train:
        assert b["point_ids"].shape == (len(locs),) and int(b["point_ids"].max()) < 9000


def test_normals_and_feature_selection(tmp_path):
    from occuseg_amd import data
    cfg = {"scale": 50, "full_scale": 4096, "use_normal": True, "use_feature": "lcnh", "regress_sigma": 0.2}
    scenes = [data.load_scene(_scene_file(tmp_path, 5, with_normals=True))]
    # default = the reference's key test ('normal' in the dict, scannet.py:90, while the files say 'normals'): d is the
    # COORDINATES, rotated and divided by the scale -- not unit vectors
    ref = data.ScanNetCollate(cfg, rng=np.random.default_rng(0)).trainMerge([0], scenes)
    assert not torch.allclose(torch.norm(ref["x"][2], dim=1), torch.ones(len(ref["x"][0])), atol=1e-2)
    b = data.ScanNetCollate(dict(cfg, reference_normal_key=False), rng=np.random.default_rng(0)).trainMerge([0], scenes)
    locs, feats, normals, bs = b["x"]
    assert feats.shape == (len(locs), 3 + 3 + 3 + 1) and normals.shape == (len(locs), 3) and bs == 1
    # normals are rotated / flipped with the scene and keep unit length (d @ m / scale, scannet.py:120)
    assert torch.allclose(torch.norm(normals, dim=1), torch.ones(len(locs)), atol=1e-4)
    assert torch.allclose(torch.norm(feats[:, :3], dim=1), torch.ones(len(locs)), atol=1e-4)   # 'l'
    assert torch.equal(feats[:, 6:9], normals) and torch.equal(feats[:, 9], locs[:, 2])       # 'n', 'h'


def test_collate_against_the_reference_trainmerge_fixture():
    """tests/golden/g7_trainmerge.npz = what the reference's OWN ScanNet.trainMerge / valMerge
    (examples/ScanNet/datasets/scannet.py:69-424, imported unchanged by tests/golden/make_g7.py in the build container)
    return for three stored scenes with elastic distortion, rotation noise, normals and 'lcnh' features.  Same scenes,
    same seeds, the reference's random sources (data.LegacyRNG): every key must agree -- integers exactly, floats to
    float32 rounding (the per-instance loops are index_add_ reductions here)."""
    import os
    from occuseg_amd import data
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_trainmerge.npz"))
    scenes = [{k: g["scene%d_%s" % (i, k)] for k in ("coords", "colors", "w", "normals", "region")} for i in range(3)]
    cfg = {"scale": 50, "full_scale": 4096, "use_normal": True, "use_elastic": True, "use_feature": "lcnh",
           "use_rotation_noise": True, "regress_sigma": 0.2}
    data.LegacyRNG.seed(100)  # ScanNet.__init__, scannet.py:42-45
    col = data.ScanNetCollate(cfg, rng=data.LegacyRNG())
    train = col.trainMerge([2, 0], scenes)
    val = col.valMerge([1, 2], scenes, list(g["arg_valOffsets"]))
    for prefix, b in (("train_", train), ("val_", val)):
        locs, feats, normals, bs = b["x"]
        assert bs == int(g[prefix + "batch"])
        want_locs = g[prefix + "locs"]
        assert locs.shape == want_locs.shape, (locs.shape, want_locs.shape)
        # voxel coordinates: the elastic field is interpolated in float64 on both sides; float32 rounding only
        assert np.abs(locs.numpy() - want_locs).max() < 2e-3, np.abs(locs.numpy() - want_locs).max()
        assert np.array_equal(locs.numpy()[:, 3], want_locs[:, 3])
        assert np.abs(feats.numpy() - g[prefix + "feats"]).max() < 1e-4
        assert np.abs(normals.numpy() - g[prefix + "normals"]).max() < 1e-5
        for k in ("y", "sizes", "regions", "region_masks", "region_indexs"):
            assert np.array_equal(b[k].numpy(), g[prefix + k]), k
        assert np.array_equal(b["instance_masks"].numpy(), g[prefix + "instance_masks"])
        assert np.abs(b["instance_sizes"].numpy() - g[prefix + "instance_sizes"]).max() < 1e-6
        assert np.abs(b["offsets"].numpy() - g[prefix + "offsets"]).max() < 1e-4
        assert np.abs(b["displacements"].numpy() - g[prefix + "displacements"]).max() < 1e-4
        assert np.array_equal(np.concatenate([m.argmax(1).numpy() for m in b["masks"]]), g[prefix + "mask_argmax"])
        assert np.array_equal(np.concatenate([m.sum(1).numpy() for m in b["masks"]]), g[prefix + "mask_rowsum"])
        assert np.array_equal(np.concatenate([i.numpy() for i in b["idxs"]]), g[prefix + "idxs"])
    assert np.array_equal(val["point_ids"].numpy(), g["val_point_ids"])
