"""-m gpu: SURVEY.md section 8 f2 -- scenes in the reference's on-disk format through the reference-shaped collate
and the prefetching loader wrapper into the GPU backend: same results as handing the same batches over by hand, the
grids and rulebooks of every batch really were built ahead (adopted, not rebuilt), float coordinates end to end."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _write_scenes(tmp_path, n):
    from occuseg_amd import scenes
    paths = []
    for i in range(n):
        vox, col = scenes.scene_points(3000 + i, target_sites=4000)
        d = {"coords": (vox.astype(np.float32) + 0.5) / 50.0, "colors": col,
             "w": np.stack([np.arange(len(vox)) % 20, (vox[:, 0] // 40) + 3], 1).astype(np.float64)}
        p = str(tmp_path / ("scene%04d_00_instance.pth" % i))
        torch.save(d, p)
        paths.append(p)
    return paths


def test_prefetching_loader_feeds_the_backend(gpu, tmp_path):
    import occuseg_amd
    from occuseg_amd import SCN, data
    scn = occuseg_amd.install()
    paths = _write_scenes(tmp_path, 4)
    scenes_ = [data.load_scene(p) for p in paths]
    cfg = {"scale": 50, "full_scale": 4096, "use_normal": False, "use_feature": "c", "regress_sigma": 0.2}
    col = data.ScanNetCollate(cfg, rng=np.random.default_rng(3))
    batches = [col.trainMerge(tbl, scenes_, paths) for tbl in ([0, 1], [2, 3], [1, 3])]
    torch.manual_seed(0)
    m = 16
    net = scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(
        scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
        scn.UNet(3, 1, [m, 2 * m, 3 * m], residual_blocks=True)).add(
        scn.BatchNormReLU(m)).add(scn.OutputLayer(3)).to(gpu)
    inp = next(mod for mod in net.modules() if isinstance(mod, scn.InputLayer))
    # by hand: float locs moved to the device, no prefetch
    plain = []
    for b in batches:
        x = [b["x"][0].to(gpu), b["x"][1].to(gpu), b["x"][2], b["x"][3]]
        plain.append(net(x).detach().clone())
    # the voxelisation of the float coordinates on the GPU == truncation on the host (sparseconvnet/ioLayers.py:56)
    x = [batches[0]["x"][0].long(), batches[0]["x"][1].to(gpu), None, 2]
    assert torch.equal(net(x), plain[0])
    # through the loader wrapper: every batch arrives on the device with its grids being built on the side stream
    adopted = []
    loader = data.PrefetchingLoader(batches, inp, gpu, move=("y", "offsets"))
    assert len(loader) == 3
    for i, b in enumerate(loader):
        assert b["x"][0].is_cuda and b["x"][0].dtype == torch.float32 and b["y"].is_cuda and b["offsets"].is_cuda
        assert id(b["x"][0]) in SCN._PREFETCHED
        out = net(b["x"])
        adopted.append(id(b["x"][0]) not in SCN._PREFETCHED)
        out.sum().backward()
        assert torch.equal(out.detach(), plain[i])
        assert out.shape[0] == b["y"].shape[0]
    assert all(adopted)
