"""-m gpu: the fork's normal-guided filters (use_normal) and cross-scale scattering against
the oracle: per-site normals and orientation classes, the remapped rulebook bit for bit in the reference's list
order, forward / dInput / dWeight of a SubmanifoldConvolution over it, ResolutionBasedScattering, upsample_feature."""
import numpy as np
import pytest
import torch

from helpers import assert_close, random_points

pytestmark = pytest.mark.gpu
F3 = torch.tensor([3, 3, 3])


def _case(seed, n=4000, batch=2, extent=24):
    rng = np.random.default_rng(seed)
    coords = random_points(seed, n, batch, extent, origin=5)
    normals = rng.normal(size=(len(coords), 3)).astype(np.float32)
    normals[::7] = 0.0                      # zero normals: normalize() leaves them alone, class 0
    normals[1::11, 1] = normals[1::11, 0]   # exact ties |x| == |y|: x wins
    return coords, normals


@pytest.mark.parametrize("path", ["sort_hash", "bitmaps"])
@pytest.mark.parametrize("seed", [0, 1])
def test_normal_guided_rulebook_bit_exact(gpu, seed, path, monkeypatch):
    """(`bitmaps`, round 6: a normal-guided batch is built from occupancy bitmaps too -- all of it but the input scale's 3x3x3
    rulebook, which the normals re-sort and getSubmanifoldRuleBook builds from the hash map it makes for that)"""
    from occuseg_amd import SCN
    from oracle import rules as R
    coords, normals = _case(seed)
    monkeypatch.setattr(SCN, "OCC_BUILD", path == "bitmaps")
    SCN._PLANS.clear()
    built = SCN.OCC_STATS["built"]
    grid, _, table, _ = R.input_layer_rules(coords)
    want_sn = R.site_normals(normals, table)
    want_ori = R.oriented_filter(want_sn)
    plain, _ = R.submanifold_rules(grid)
    want = R.remap_rules_with_normal(plain, want_ori)
    m = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    m.inputLayer(size, torch.from_numpy(coords), 0, 4, gpu, normals=torch.from_numpy(normals).to(gpu))
    assert SCN.OCC_STATS["built"] == built + (1 if path == "bitmaps" else 0)
    sn, ori = m.normals[(4096, 4096, 4096)]
    assert np.array_equal(ori.cpu().numpy().astype(np.int64), want_ori)
    assert set(np.unique(want_ori)) == {0, 2, 4}
    assert np.allclose(sn.cpu().numpy(), want_sn, rtol=0, atol=1e-6)
    rb = m.getSubmanifoldRuleBook(size, F3)
    got = rb.host_lists()
    for k in range(27):
        assert np.array_equal(got[k], want[k]), k
    # the device copy is the same set of pairs per list, sorted by output site
    dev = rb.rules.cpu().numpy()
    cnt = rb.counts.cpu().tolist()
    for k in range(27):
        d = dev[k, :cnt[k]]
        assert np.all(np.diff(d[:, 1]) > 0)
        assert np.array_equal(d[np.lexsort((d[:, 0], d[:, 1]))], want[k][np.lexsort((want[k][:, 0], want[k][:, 1]))])
    # strided rules at a scale the normals do not guide (default normal_guide_scale 10240 > 4096): plain fast path
    rb2 = m.getRuleBook(size, size // 2, torch.tensor([2, 2, 2]), torch.tensor([2, 2, 2]))
    _, want2, _ = R.strided_rules(grid)
    for k in range(8):
        assert np.array_equal(rb2.host_lists()[k], want2[k])
    assert (2048, 2048, 2048) not in m.normals
    # ... and refused where the reference would switch to its hash-map path
    m2 = SCN.Metadata_3()
    m2.setNormalGuideScale(4096)
    m2.inputLayer(size, torch.from_numpy(coords), 0, 4, gpu, normals=torch.from_numpy(normals).to(gpu))
    with pytest.raises(NotImplementedError, match="normal-guided STRIDED"):
        m2.getRuleBook(size, size // 2, torch.tensor([2, 2, 2]), torch.tensor([2, 2, 2]))


@pytest.mark.parametrize("ci,co", [(16, 32), (64, 64)])
def test_normal_guided_convolution_fwd_bwd(gpu, ci, co):
    """the operator API end to end: InputLayer([coords, feats, normals, B]) -> SubmanifoldConvolution, autograd"""
    import occuseg_amd
    from oracle import arith as A, rules as R
    scn = occuseg_amd.install()
    coords, normals = _case(3, n=3000)
    grid, _, table, _ = R.input_layer_rules(coords)
    ori = R.oriented_filter(R.site_normals(normals, table))
    plain, _ = R.submanifold_rules(grid)
    rules = R.remap_rules_with_normal(plain, ori)
    rng = np.random.default_rng(ci)
    feats = rng.normal(size=(len(coords), ci)).astype(np.float32)
    torch.manual_seed(ci + co)
    inp = scn.InputLayer(3, 4096, mode=4)
    conv = scn.SubmanifoldConvolution(3, ci, co, 3, True).to(gpu)
    with torch.no_grad():
        conv.bias.normal_()
    f = torch.from_numpy(feats).to(gpu).requires_grad_(True)
    t = inp([torch.from_numpy(coords), f, torch.from_numpy(normals).to(gpu), 2])
    xs = t.features.detach().cpu().numpy()
    out = conv(t).features
    w = conv.weight.detach().cpu().numpy()
    b = conv.bias.detach().cpu().numpy()
    want, _ = A.conv_forward(xs, w, rules, grid.n, b, dtype=np.float64)
    assert_close(out.detach().cpu().numpy(), want, "normal-guided forward")
    dy = rng.normal(size=want.shape).astype(np.float32)
    xt = t.features
    xt.retain_grad()
    out.backward(torch.from_numpy(dy).to(gpu))
    g_x = xt.grad
    d_in, d_w, d_b = A.conv_backward(xs, dy, w, rules, want_bias=True, dtype=np.float64)
    assert_close(g_x.cpu().numpy(), d_in, "normal-guided dInput")
    assert_close(conv.weight.grad.cpu().numpy(), d_w, "normal-guided dWeight")
    assert_close(conv.bias.grad.cpu().numpy(), d_b, "normal-guided dBias")
    # without normals the same module gives the plain convolution (the remap really changed something)
    t0 = inp([torch.from_numpy(coords), f.detach(), torch.zeros(3, 3), 2])
    plain_out, _ = A.conv_forward(xs, w, plain, grid.n, b, dtype=np.float64)
    assert_close(conv(t0).features.detach().cpu().numpy(), plain_out, "plain forward")
    assert np.abs(plain_out - want).max() > 1e-2 * np.abs(want).max()


def test_resolution_based_scattering_and_upsample(gpu):
    import occuseg_amd
    from occuseg_amd import SCN
    from oracle import rules as R
    scn = occuseg_amd.install()
    coords = random_points(5, 3000, 3, 40, origin=3)
    rng = np.random.default_rng(5)
    feats = rng.normal(size=(len(coords), 8)).astype(np.float32)
    net = scn.Sequential().add(scn.InputLayer(3, 4096, mode=4))
    hr = net([torch.from_numpy(coords), torch.from_numpy(feats).to(gpu), None, 3])
    down = scn.Sequential().add(scn.Convolution(3, 8, 16, 2, 2, False)).add(scn.Convolution(3, 16, 16, 2, 2, False)).to(gpu)
    lr = down(hr)  # two levels down: stride 4
    loc_hr = hr.get_spatial_locations().numpy()
    loc_lr = lr.get_spatial_locations().numpy()
    # the boundary function, one sample at a time as sparseconvnet/utils.py:85-90 calls it
    start = 0
    for k in range(3):
        plr, phr = loc_lr[loc_lr[:, 3] == k, :3], loc_hr[loc_hr[:, 3] == k, :3]
        got = SCN.ResolutionBasedScattering(lr.metadata, torch.from_numpy(plr).int().to(gpu),
                                            torch.from_numpy(phr).int().to(gpu), 4)
        want = R.resolution_based_scattering(plr, phr, 4)
        assert np.array_equal(got.cpu().numpy(), want) and want.min() >= 0
        assert np.array_equal(plr[want], phr // 4)
        start += len(plr)
    # a point whose parent is missing
    got = SCN.ResolutionBasedScattering(lr.metadata, torch.tensor([[1, 1, 1]], dtype=torch.int32, device=gpu),
                                        torch.tensor([[4, 4, 4], [9, 9, 9]], dtype=torch.int32, device=gpu), 4)
    assert got.cpu().tolist() == [0, -1]
    up = scn.upsample_feature(lr=lr, hr=hr, stride=4)
    # every fine site gets the feature row of the coarse site above it
    gl = {tuple(r): i for i, r in enumerate(map(tuple, loc_lr))}
    idx = np.array([gl[(x // 4, y // 4, z // 4, b)] for x, y, z, b in loc_hr])
    assert torch.equal(up.features, lr.features[torch.from_numpy(idx).to(gpu)])
    assert up.spatial_size is hr.spatial_size and up.metadata is hr.metadata
    # bilinear variant: weights sum to one, so a constant field stays constant
    lr.features = torch.ones_like(lr.features)
    upb = scn.upsample_feature(lr=lr, hr=hr, stride=4, bilinear=True)
    ok = torch.isfinite(upb.features).all(1)
    assert ok.float().mean() > 0.9 and torch.allclose(upb.features[ok], torch.ones_like(upb.features[ok]), atol=1e-5)
