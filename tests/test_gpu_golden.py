"""-m gpu: the HIP path against the committed golden vectors (tests/golden/*.npz: rulebooks from the
oracle, floating-point expectations from the reference's own CPU loops, see make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_close, build_metadata

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F3, F2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


@pytest.mark.parametrize("name", ["g1_rules_8cube", "g2_rules_bernoulli32", "g3_rules_bernoulli14"])
def test_golden_rulebooks_bit_exact(gpu, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    m, size = build_metadata(g["coords"], gpu)
    assert np.array_equal(m.getSpatialLocations(size).numpy().astype(np.int32), g["site_coords"])
    h, t = m.getInputRulesHost()
    assert np.array_equal(h, g["header"]) and np.array_equal(t, g["table"])
    assert np.array_equal(m.input_site_of_row.cpu().numpy(), g["site_of_point"])
    got = m.getSubmanifoldRuleBook(size, F3).host_lists()
    assert all(np.array_equal(got[k], g["sub%02d" % k]) for k in range(27))
    got = m.getRuleBook(size, size // 2, F2, F2).host_lists()
    assert all(np.array_equal(got[k], g["str%d" % k]) for k in range(8))
    assert np.array_equal(m.getSpatialLocations(size // 2).numpy().astype(np.int32), g["coarse_coords"])


def test_golden_conv_deconv(gpu):
    from occuseg_amd import SCN
    g = np.load(os.path.join(GOLD, "g3_conv_bernoulli14.npz"))
    r = np.load(os.path.join(GOLD, "g3_rules_bernoulli14.npz"))
    m, size = build_metadata(r["coords"], gpu)
    half = size // 2
    e = torch.empty(0, device=gpu)
    out = torch.empty(0, device=gpu)
    SCN.SubmanifoldConvolution_updateOutput(size, F3, m, _t(g["x"], gpu), out, _t(g["w"], gpu), e, 1)
    assert_close(out.cpu().numpy(), g["y"], "SC fwd")
    gi, gw = torch.empty(0, device=gpu), torch.zeros(27, 16, 16, device=gpu)
    SCN.SubmanifoldConvolution_backward(size, F3, m, _t(g["x"], gpu), gi, _t(g["dy"], gpu), _t(g["w"], gpu), gw, e, 1)
    assert_close(gi.cpu().numpy(), g["dx"], "SC dInput")
    assert_close(gw.cpu().numpy(), g["dw"], "SC dWeight")
    out = torch.empty(0, device=gpu)
    SCN.Convolution_updateOutput(size, half, F2, F2, m, _t(g["x"], gpu), out, _t(g["wc"], gpu), e)
    assert_close(out.cpu().numpy(), g["yc"], "conv fwd")
    gi, gw = torch.empty(0, device=gpu), torch.zeros(8, 16, 32, device=gpu)
    SCN.Convolution_backward(size, half, F2, F2, m, _t(g["x"], gpu), gi, _t(g["dyc"], gpu), _t(g["wc"], gpu), gw, e)
    assert_close(gi.cpu().numpy(), g["dxc"], "conv dInput")
    assert_close(gw.cpu().numpy(), g["dwc"], "conv dWeight")
    out = torch.empty(0, device=gpu)
    SCN.Deconvolution_updateOutput(half, size, F2, F2, m, _t(g["xd"], gpu), out, _t(g["wd"], gpu), e)
    assert_close(out.cpu().numpy(), g["yd"], "deconv fwd")
    gi, gw = torch.empty(0, device=gpu), torch.zeros(8, 32, 16, device=gpu)
    SCN.Deconvolution_backward(half, size, F2, F2, m, _t(g["xd"], gpu), gi, _t(g["dyd"], gpu), _t(g["wd"], gpu), gw, e)
    assert_close(gi.cpu().numpy(), g["dxd"], "deconv dInput")
    assert_close(gw.cpu().numpy(), g["dwd"], "deconv dWeight")


def test_golden_batchnorm_and_io(gpu):
    from occuseg_amd import SCN
    g = np.load(os.path.join(GOLD, "g4_batchnorm.npz"))
    for nn in (1024, 2):
        x, gm, b, rm0, rv0, dy = (g["%s%d" % (k, nn)] for k in ("x", "g", "b", "rm", "rv", "dy"))
        for li, leak in enumerate((0.0, 0.333, 1.0)):
            for train in (1, 0):
                tag = "%d_l%d_t%d" % (nn, li, train)
                out, sm, si = (torch.empty(0, device=gpu) for _ in range(3))
                rm, rv = _t(rm0, gpu), _t(rv0, gpu)
                SCN.BatchNormalization_updateOutput(_t(x, gpu), out, sm, si, rm, rv, _t(gm, gpu), _t(b, gpu), 1e-4,
                                                    0.9, bool(train), leak)
                assert_close(out.cpu().numpy(), g["y" + tag], "bn y " + tag)
                assert_close(rm.cpu().numpy(), g["rmo" + tag], "bn running_mean")
                assert_close(rv.cpu().numpy(), g["rvo" + tag], "bn running_var")
                if train:
                    gi, gg, gb = torch.empty(0, device=gpu), torch.zeros(16, device=gpu), torch.zeros(16, device=gpu)
                    dyt = _t(dy, gpu)
                    # the golden y is the mask input, exactly as the reference's backward consumes output_features
                    SCN.BatchNormalization_backward(_t(x, gpu), gi, _t(g["y" + tag], gpu), dyt, _t(g["sm" + tag], gpu),
                                                    _t(g["si" + tag], gpu), rm, rv, _t(gm, gpu), _t(b, gpu), gg, gb, leak)
                    scale = max(float(np.abs(g["dm" + tag] * g["si" + tag] * gm).max()),
                                float(np.abs(g["dx" + tag]).max()))
                    assert_close(gi.cpu().numpy(), g["dx" + tag], "bn dInput", scale=scale)
                    assert_close(gg.cpu().numpy(), g["dw" + tag], "bn dWeight")
                    assert_close(gb.cpu().numpy(), g["db" + tag], "bn dBias")
                    assert_close(dyt.cpu().numpy(), g["dm" + tag], "bn masked dOut", 1e-6)
    io = np.load(os.path.join(GOLD, "g5_io_layers.npz"))
    coords = np.load(os.path.join(GOLD, "g1_rules_8cube.npz"))["coords"]
    for mode in (3, 4):
        m = SCN.Metadata_3()
        out = torch.empty(0, device=gpu)
        SCN.InputLayer_updateOutput(m, torch.tensor([4096] * 3), torch.from_numpy(coords), _t(io["feats"], gpu), out, 0,
                                    mode, None)
        assert_close(out.cpu().numpy(), io["out_m%d" % mode], "input fwd")
        gi = torch.empty(0, device=gpu)
        SCN.InputLayer_updateGradInput(m, gi, _t(io["out_m%d" % mode], gpu))
        assert_close(gi.cpu().numpy(), io["din_m%d" % mode], "input bwd")
