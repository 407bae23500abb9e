"""-m gpu: gather-mm-scatter, batch norm and I/O layers through the SCN boundary vs the float64 oracle."""
import numpy as np
import pytest
import torch

from helpers import assert_close, build_metadata, random_points

pytestmark = pytest.mark.gpu

F3 = torch.tensor([3, 3, 3])
F2 = torch.tensor([2, 2, 2])


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def _setup(gpu, seed=0, n=3000, batch=2, extent=24):
    from oracle import rules as R
    coords = random_points(seed, n, batch, extent, origin=3)
    m, size = build_metadata(coords, gpu)
    grid, header, table, sop = R.input_layer_rules(coords)
    return m, size, grid, table, coords


# (Ci, Co): MFMA tile path (multiples of 16, NW = 4/3/2/1), Ci chunking (>64), generic path (5, 20; 3 -> 8), the stem on the
# matrix cores (k_conv_stem / k_dw_stem: Ci <= 4, Co % 16 == 0; one, two and one-and-a-half 64-plane blocks)
PLANES = [(16, 16), (32, 64), (64, 64), (64, 48), (48, 32), (128, 64), (80, 16), (3, 32), (5, 7), (20, 24),
          (1, 16), (2, 96), (3, 64), (4, 80), (4, 128), (3, 8)]


@pytest.mark.parametrize("ci,co", PLANES)
def test_submanifold_fwd_bwd(gpu, ci, co):
    from occuseg_amd import SCN
    from oracle import arith as A, rules as R
    m, size, grid, _, _ = _setup(gpu)
    rules, _ = R.submanifold_rules(grid)
    rng = np.random.default_rng(ci * 100 + co)
    x = rng.normal(size=(grid.n, ci)).astype(np.float32)
    w = (rng.normal(size=(27, ci, co)) * (2.0 / ci / 27) ** 0.5).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    dy = rng.normal(size=(grid.n, co)).astype(np.float32)
    for bias in (None, b):
        want, macs = A.conv_forward(x, w, rules, grid.n, bias, dtype=np.float64)
        out = torch.empty(0, device=gpu)
        bt = _t(bias, gpu) if bias is not None else torch.empty(0, device=gpu)
        got_macs = SCN.SubmanifoldConvolution_updateOutput(size, F3, m, _t(x, gpu), out, _t(w, gpu), bt, 1)
        assert got_macs == macs
        assert_close(out.cpu().numpy(), want, "submanifold fwd %d->%d" % (ci, co))
    d_in, d_w, d_b = A.conv_backward(x, dy, w, rules, want_bias=True, dtype=np.float64)
    gi = torch.empty(0, device=gpu)
    gw = torch.zeros(27, ci, co, device=gpu)
    gb = torch.zeros(co, device=gpu)
    SCN.SubmanifoldConvolution_backward(size, F3, m, _t(x, gpu), gi, _t(dy, gpu), _t(w, gpu), gw, gb, 1)
    assert_close(gi.cpu().numpy(), d_in, "submanifold dInput")
    assert_close(gw.cpu().numpy(), d_w, "submanifold dWeight")
    assert_close(gb.cpu().numpy(), d_b, "submanifold dBias")


# the last two run on a grid large enough (>= 512 output tiles on both scales) for the two-wave kernel k_conv_tile4
@pytest.mark.parametrize("ci,co,big", [(16, 32, 0), (64, 128, 0), (96, 64, 0), (32, 48, 0), (3, 8, 0), (128, 192, 0),
                                       (3, 32, 0), (4, 96, 0), (64, 64, 1), (64, 128, 1)])
def test_conv_deconv_fwd_bwd(gpu, ci, co, big):
    from occuseg_amd import SCN
    from oracle import arith as A, rules as R
    m, size, grid, _, _ = _setup(gpu, seed=1, n=100000, batch=2, extent=70) if big else _setup(gpu, seed=1)
    coarse, rules, _ = R.strided_rules(grid)
    assert not big or coarse.n >= 512 * 128
    out_size = size // 2
    rng = np.random.default_rng(ci + co)
    x = rng.normal(size=(grid.n, ci)).astype(np.float32)
    w = (rng.normal(size=(8, ci, co)) * (2.0 / ci / 8) ** 0.5).astype(np.float32)
    dy = rng.normal(size=(coarse.n, co)).astype(np.float32)
    empty = torch.empty(0, device=gpu)
    # strided convolution: fine -> coarse
    want, macs = A.conv_forward(x, w, rules, coarse.n, dtype=np.float64)
    out = torch.empty(0, device=gpu)
    got_macs = SCN.Convolution_updateOutput(size, out_size, F2, F2, m, _t(x, gpu), out, _t(w, gpu), empty)
    assert got_macs == macs and out.shape == (coarse.n, co)
    assert_close(out.cpu().numpy(), want, "conv fwd")
    d_in, d_w, _ = A.conv_backward(x, dy, w, rules, dtype=np.float64)
    gi, gw = torch.empty(0, device=gpu), torch.zeros(8, ci, co, device=gpu)
    SCN.Convolution_backward(size, out_size, F2, F2, m, _t(x, gpu), gi, _t(dy, gpu), _t(w, gpu), gw, empty)
    assert_close(gi.cpu().numpy(), d_in, "conv dInput")
    assert_close(gw.cpu().numpy(), d_w, "conv dWeight")
    # deconvolution: coarse -> fine, own weights [8, co, ci], same rulebook with columns swapped
    xd = rng.normal(size=(coarse.n, co)).astype(np.float32)
    wd = (rng.normal(size=(8, co, ci)) * (2.0 / co / 8) ** 0.5).astype(np.float32)
    dyd = rng.normal(size=(grid.n, ci)).astype(np.float32)
    want, macs = A.conv_forward(xd, wd, rules, grid.n, in_col=1, out_col=0, dtype=np.float64)
    out = torch.empty(0, device=gpu)
    SCN.Deconvolution_updateOutput(out_size, size, F2, F2, m, _t(xd, gpu), out, _t(wd, gpu), empty)
    assert_close(out.cpu().numpy(), want, "deconv fwd")
    d_in, d_w, _ = A.conv_backward(xd, dyd, wd, rules, in_col=1, out_col=0, dtype=np.float64)
    gi, gw = torch.empty(0, device=gpu), torch.zeros(8, co, ci, device=gpu)
    SCN.Deconvolution_backward(out_size, size, F2, F2, m, _t(xd, gpu), gi, _t(dyd, gpu), _t(wd, gpu), gw, empty)
    assert_close(gi.cpu().numpy(), d_in, "deconv dInput")
    assert_close(gw.cpu().numpy(), d_w, "deconv dWeight")


@pytest.mark.parametrize("n,c", [(8192, 16), (5000, 64), (2, 16), (70001, 96), (333, 7), (1200, 768)])
@pytest.mark.parametrize("leak", [0.0, 0.333, 1.0])
def test_batchnorm(gpu, n, c, leak):
    from occuseg_amd import SCN
    from oracle import arith as A
    rng = np.random.default_rng(n + c)
    x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 2, c) + rng.normal(size=c)).astype(np.float32)
    g = rng.uniform(0.5, 1.5, c).astype(np.float32)
    b = rng.normal(size=c).astype(np.float32)
    rm0 = rng.normal(size=c).astype(np.float32)
    rv0 = rng.uniform(0.5, 2, c).astype(np.float32)
    dy = rng.normal(size=(n, c)).astype(np.float32)
    for train in (True, False):
        y, mean, invstd, rm, rv = A.batchnorm_forward(x, g, b, rm0, rv0, 1e-4, 0.9, train, leak, dtype=np.float64)
        out, sm, si = torch.empty(0, device=gpu), torch.empty(0, device=gpu), torch.empty(0, device=gpu)
        rmt, rvt = _t(rm0, gpu), _t(rv0, gpu)
        SCN.BatchNormalization_updateOutput(_t(x, gpu), out, sm, si, rmt, rvt, _t(g, gpu), _t(b, gpu), 1e-4, 0.9,
                                            train, leak)
        assert_close(out.cpu().numpy(), y, "bn y train=%s" % train)
        assert_close(sm.cpu().numpy(), mean, "bn saveMean")
        assert_close(si.cpu().numpy(), invstd, "bn saveInvStd")
        assert_close(rmt.cpu().numpy(), rm, "bn running_mean")
        assert_close(rvt.cpu().numpy(), rv, "bn running_var")
        if train:
            # backward takes output_features as an INPUT (the ReLU mask is y > 0): give the oracle the y the
            # device produced, otherwise elements with |y| ~ 1e-8 flip the mask and differ by a whole d_out
            d_in, d_masked, d_g, d_b = A.batchnorm_backward(x, out.cpu().numpy(), dy, mean, invstd, g, leak,
                                                            dtype=np.float64)
            gi, gg, gb = torch.empty(0, device=gpu), torch.zeros(c, device=gpu), torch.zeros(c, device=gpu)
            dyt = _t(dy, gpu)
            SCN.BatchNormalization_backward(_t(x, gpu), gi, out, dyt, sm, si, rmt, rvt, _t(g, gpu), _t(b, gpu),
                                            gg, gb, leak)
            # dInput = (d - mean(d) - (x-mu)k) * invstd * gamma cancels heavily when N is tiny (for N=2 it is
            # analytically ~0): hold it to 1e-4 of the magnitude of its terms, not of the cancelled result
            term = float(np.abs(d_masked * invstd * g).max())
            assert_close(gi.cpu().numpy(), d_in, "bn dInput", scale=max(term, float(np.abs(d_in).max())))
            assert_close(gg.cpu().numpy(), d_g, "bn dWeight")
            assert_close(gb.cpu().numpy(), d_b, "bn dBias")
            # the reference contract: d_output_features is rewritten in place (BatchNormalization.cu:151-153)
            assert_close(dyt.cpu().numpy(), d_masked, "bn masked dOut written back")
            # the autograd module's mode: same results, d_output_features untouched
            dyt2, gi2 = _t(dy, gpu), torch.empty(0, device=gpu)
            gg2, gb2 = torch.zeros(c, device=gpu), torch.zeros(c, device=gpu)
            SCN.BatchNormalization_backward(_t(x, gpu), gi2, out, dyt2, sm, si, rmt, rvt, _t(g, gpu), _t(b, gpu),
                                            gg2, gb2, leak, _rewrite_d_output=False)
            assert torch.equal(gi2, gi) and torch.equal(gg2, gg) and torch.equal(gb2, gb)
            assert np.array_equal(dyt2.cpu().numpy(), dy)
            # ... and with the mask recomputed from x instead of read from y (SCN_BN_MASK_FROM_X; up to 2048 rows this
            # is the single-launch kernel): y is not even passed, d_output_features untouched, accumulate variant too
            dyt3, gi3 = _t(dy, gpu), torch.empty(0, device=gpu)
            gg3, gb3 = torch.ones(c, device=gpu), torch.ones(c, device=gpu)
            SCN.BatchNormalization_backward(_t(x, gpu), gi3, torch.empty(0, device=gpu), dyt3, sm, si, rmt, rvt,
                                            _t(g, gpu), _t(b, gpu), gg3, gb3, leak, _mask_from_x=True, _accumulate=True)
            assert_close(gi3.cpu().numpy(), d_in, "bn dInput (mask from x)", scale=max(term, float(np.abs(d_in).max())))
            assert_close(gg3.cpu().numpy() - 1, d_g, "bn dWeight (mask from x, accumulated)",
                         scale=max(1.0, float(np.abs(d_g).max())))
            assert_close(gb3.cpu().numpy() - 1, d_b, "bn dBias (mask from x, accumulated)",
                         scale=max(1.0, float(np.abs(d_b).max())))
            assert np.array_equal(dyt3.cpu().numpy(), dy)


def test_batchnorm_no_affine(gpu):
    from occuseg_amd import SCN
    from oracle import arith as A
    rng = np.random.default_rng(5)
    x = rng.normal(size=(900, 32)).astype(np.float32)
    y, mean, invstd, rm, rv = A.batchnorm_forward(x, None, None, np.zeros(32), np.ones(32), dtype=np.float64)
    e = torch.empty(0, device=gpu)
    out, sm, si = torch.empty(0, device=gpu), torch.empty(0, device=gpu), torch.empty(0, device=gpu)
    SCN.BatchNormalization_updateOutput(_t(x, gpu), out, sm, si, torch.zeros(32, device=gpu),
                                        torch.ones(32, device=gpu), e, e, 1e-4, 0.9, True, 1.0)
    assert_close(out.cpu().numpy(), y, "bn no affine")


@pytest.mark.parametrize("mode", [3, 4])
@pytest.mark.parametrize("c", [3, 16])
def test_io_layers(gpu, mode, c):
    from occuseg_amd import SCN
    from oracle import arith as A, rules as R
    coords = random_points(3, 2000, 3, 12)
    grid, header, table, sop = R.input_layer_rules(coords, mode=mode)
    rng = np.random.default_rng(mode * 10 + c)
    feats = rng.normal(size=(len(coords), c)).astype(np.float32)
    m = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    out = torch.empty(0, device=gpu)
    SCN.InputLayer_updateOutput(m, size, torch.from_numpy(coords), _t(feats, gpu), out, 0, mode, None)
    assert_close(out.cpu().numpy(), A.input_layer_forward(feats, table, mode, np.float64), "input fwd")
    d_out = rng.normal(size=(grid.n, c)).astype(np.float32)
    gi = torch.empty(0, device=gpu)
    SCN.InputLayer_updateGradInput(m, gi, _t(d_out, gpu))
    assert_close(gi.cpu().numpy(), A.input_layer_backward(d_out, table, len(coords), mode, np.float64), "input bwd")
    # output layer: site row copied to every original point; backward sums them back
    o = torch.empty(0, device=gpu)
    SCN.OutputLayer_updateOutput(m, _t(d_out, gpu), o)
    assert np.array_equal(o.cpu().numpy(), A.output_layer_forward(d_out, table, len(coords)))
    g2 = torch.empty(0, device=gpu)
    SCN.OutputLayer_updateGradInput(m, g2, _t(feats, gpu))
    assert_close(g2.cpu().numpy(), A.output_layer_backward(feats, table, np.float64), "output bwd")


def test_cpu_tensors_are_refused():
    """no CPU fallback: the product path fails loudly instead of computing on the host"""
    from occuseg_amd import SCN
    x = torch.zeros(4, 16)
    with pytest.raises(RuntimeError):
        SCN.BatchNormalization_updateOutput(x, torch.empty(0), torch.empty(0), torch.empty(0), torch.zeros(16),
                                            torch.ones(16), torch.empty(0), torch.empty(0), 1e-4, 0.9, True, 1.0)


@pytest.mark.parametrize("ci,co,with_bias", [(128, 64, False), (256, 128, True), (192, 64, True), (64, 32, False),
                                             (20, 12, True)])
def test_network_in_network(gpu, ci, co, with_bias):
    """1x1 convolution: the one-contribution-per-row kernel over the identity rulebook (planes % 64 == 0: k_conv_expand;
    % 32: the tile kernels) or a library GEMM; with and without bias; 1, 3 and 4 chunks of C_i"""
    from occuseg_amd import SCN
    from oracle import arith as A
    rng = np.random.default_rng(ci + co)
    n = 3001
    x = rng.normal(size=(n, ci)).astype(np.float32)
    w = (rng.normal(size=(ci, co)) * (2.0 / ci) ** 0.5).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32) if with_bias else None
    dy = rng.normal(size=(n, co)).astype(np.float32)
    e = torch.empty(0, device=gpu)
    out = torch.empty(0, device=gpu)
    macs = SCN.NetworkInNetwork_updateOutput(_t(x, gpu), out, _t(w, gpu), _t(b, gpu) if with_bias else e)
    assert macs == float(n) * ci * co
    assert_close(out.cpu().numpy(), A.nin_forward(x, w, b, dtype=np.float64), "NiN fwd")
    d_in, d_w, _ = A.nin_backward(x, dy, w, dtype=np.float64)
    gi, gw = torch.empty(0, device=gpu), torch.full((ci, co), 7.0, device=gpu)
    SCN.NetworkInNetwork_updateGradInput(gi, _t(dy, gpu), _t(w, gpu))
    SCN.NetworkInNetwork_accGradParameters(_t(x, gpu), _t(dy, gpu), gw, e)
    assert_close(gi.cpu().numpy(), d_in, "NiN dInput")
    assert_close(gw.cpu().numpy(), d_w, "NiN dWeight")


# (C_i, C_o, points per sample, extent): the last five are the shapes that carry k_conv_tile4's time in the m64 train step
# (levels 1 and 2 of the UNet, forward and -- with the planes swapped -- dInput), on grids just large enough for
# >= 512 tiles of 128 rows x 64 planes
TILE4_SHAPES = [(64, 64, 72000, 60), (128, 64, 72000, 60), (64, 128, 72000, 60), (320, 64, 72000, 60),
                (128, 128, 20000, 40), (256, 128, 20000, 40), (192, 192, 13000, 35), (384, 192, 13000, 35),
                (128, 256, 10000, 32)]


@pytest.mark.parametrize("ci,co,n,extent", TILE4_SHAPES)
def test_submanifold_forward_variants(gpu, ci, co, n, extent):
    """The barrier-free two-wave kernel k_conv_tile4 (default where it applies: >= 512 tiles, C_i a multiple of 64 up to
    384, C_o a multiple of 64) and the LDS-staged kernel (scn_set_tuning(0, 2), which only libscn_hip_tuning.so
    exports) against the float64 oracle (arithmetic of sparseconvnet/SCN/CPU/Convolution.cpp:114-189) and against
    each other; then the product library against the tuning build's default (same sources, same kernel:
    bit-identical); then dInput and dWeight of the same layer."""
    from occuseg_amd import SCN, _lib
    from oracle import arith as A, rules as R
    m, size, grid, _, _ = _setup(gpu, seed=11, n=n, batch=2, extent=extent)
    lib0 = _lib.load()
    assert lib0.scn_conv_plan(27, grid.n, grid.n, grid.n, ci, co, 0) == 401, "forward not on k_conv_tile4"
    rules, _ = R.submanifold_rules(grid)
    rng = np.random.default_rng(ci + co)
    x = rng.normal(size=(grid.n, ci)).astype(np.float32)
    w = (rng.normal(size=(27, ci, co)) * (2.0 / ci / 27) ** 0.5).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    want, macs = A.conv_forward(x, w, rules, grid.n, b, dtype=np.float64)
    assert not hasattr(_lib.load(), "scn_set_tuning")  # the product library has no tuning entry points
    outs = {}
    xt, wt, bt = _t(x, gpu), _t(w, gpu), _t(b, gpu)
    try:
        lib = _lib.use_tuning_build()
        for variant in (4, 2):
            lib.scn_set_tuning(0, variant)
            out = torch.empty(0, device=gpu)
            got = SCN.SubmanifoldConvolution_updateOutput(size, F3, m, xt, out, wt, bt, 1)
            assert got == macs
            outs[variant] = out.cpu().numpy()
        lib.scn_set_tuning(0, 4)
        # ... and k_conv_tile4 with hipcc's pass loop instead of the hand-written one (tools/gen_tile4_loop.py): bit-identical
        lib.scn_set_tuning(28, 0)
        out = torch.empty(0, device=gpu)
        SCN.SubmanifoldConvolution_updateOutput(size, F3, m, xt, out, wt, bt, 1)
        outs["c++ loop"] = out.cpu().numpy()
        lib.scn_set_tuning(28, 1)
    finally:
        _lib.use_product_build()
    out = torch.empty(0, device=gpu)
    SCN.SubmanifoldConvolution_updateOutput(size, F3, m, xt, out, wt, bt, 1)
    assert np.array_equal(out.cpu().numpy(), outs[4])
    assert np.array_equal(outs["c++ loop"], outs[4]), "the assembly pass loop and hipcc's differ"
    assert_close(outs[4], want, "variant 4 forward %d->%d" % (ci, co))
    assert_close(outs[2], want, "variant 2 forward %d->%d" % (ci, co))
    # same per-(offset, chunk) summation order in both kernels: bit-identical
    assert np.array_equal(outs[4], outs[2])
    # dInput (transposed weights, offsets reversed: planes swapped) and dWeight of the same layer
    dy = rng.normal(size=(grid.n, co)).astype(np.float32)
    d_in, d_w, _ = A.conv_backward(x, dy, w, rules, dtype=np.float64)
    gi, gw, gb = torch.empty(0, device=gpu), torch.zeros(27, ci, co, device=gpu), torch.zeros(co, device=gpu)
    SCN.SubmanifoldConvolution_backward(size, F3, m, _t(x, gpu), gi, _t(dy, gpu), _t(w, gpu), gw, gb, 1)
    assert_close(gi.cpu().numpy(), d_in, "large submanifold dInput %d->%d" % (co, ci))
    assert_close(gw.cpu().numpy(), d_w, "large submanifold dWeight")
    if (ci, co) in ((128, 128), (256, 128), (192, 192), (384, 192)):  # C_o of the dInput = ci: enough plane blocks
        assert lib0.scn_conv_plan(27, grid.n, grid.n, grid.n, co, ci, 0) == 401, "dInput not on k_conv_tile4"


def test_tile4_residual_and_column_statistics_at_384_192(gpu):
    """the widest k_conv_tile4 launch of the m64 step (decoder level 2: six 64-plane chunks, LDS to the byte) with the
    shortcut added inside the kernel and the column statistics for the following batch norm taken from the tile"""
    from occuseg_amd import SCN, _lib
    from oracle import arith as A, rules as R
    ci, co = 384, 192
    m, size, grid, _, _ = _setup(gpu, seed=12, n=13000, batch=2, extent=35)
    assert _lib.load().scn_conv_plan(27, grid.n, grid.n, grid.n, ci, co, 0) == 401
    rules, _ = R.submanifold_rules(grid)
    rng = np.random.default_rng(5)
    x = rng.normal(size=(grid.n, ci)).astype(np.float32)
    w = (rng.normal(size=(27, ci, co)) * (2.0 / ci / 27) ** 0.5).astype(np.float32)
    res = rng.normal(size=(grid.n, co)).astype(np.float32)
    want, _ = A.conv_forward(x, w, rules, grid.n, None, dtype=np.float64)
    want = want + res
    out, sink = torch.empty(0, device=gpu), []
    SCN.SubmanifoldConvolution_updateOutput(size, F3, m, _t(x, gpu), out, _t(w, gpu), torch.empty(0, device=gpu), 1,
                                            residual=_t(res, gpu), stats_out=sink)
    assert_close(out.cpu().numpy(), want, "tile4 384->192 + residual")
    assert len(sink) == 1 and sink[0].shape[1:] == (2, co)
    part = sink[0].double().sum(0).cpu().numpy()
    assert_close(part[0], want.sum(0), "column sums", scale=np.abs(want).sum(0).max())
    assert_close(part[1], (want * want).sum(0), "column sums of squares")


@pytest.mark.parametrize("ci,co,leak", [(128, 128, 0.0), (64, 256, 0.3)])
def test_stream_k_pieces_equal_one_workgroup_per_tile(gpu, ci, co, leak):
    """k_conv_tile4 as a stream-K launch (scn_conv_forward's tile_passes + sk_state): 637 tiles x 2 column blocks = 1.24 rounds of
    the 1,024 persistent workgroups -- every slice boundary cuts a tile, whose second piece starts from the slab the first
    piece published -- and x 4 column blocks = 2.49 rounds (whole tiles first, then the sliced part).  Forward with bias,
    shortcut and column statistics, then the dInput launch with the batch-norm-backward sums: against the float64 oracle, and
    BITWISE against the one-workgroup-per-tile launch and against k_conv_tile2 (same summation order in all three); the control
    block is zero again after every launch."""
    from occuseg_amd import SCN, _lib
    from oracle import arith as A, rules as R
    before = SCN.STREAM_K
    SCN.STREAM_K = True  # (rulebooks carry their pass prefix and slice tables only when it is on: SCN_STREAM_K=0 must not break the test)
    try:
        m, size, grid, _, _ = _setup(gpu, seed=21, n=45000, batch=2, extent=60)
        m.getSubmanifoldRuleBook(size, F3)
    finally:
        SCN.STREAM_K = before
    N = grid.n
    lib0 = _lib.load()
    tiles = (N + 127) // 128
    assert lib0.scn_conv_plan(27, N, N, N, ci, co, 0) == 401 and tiles == 637  # x 2 = 1.24 rounds, x 4 = 2.49 rounds
    # the product library's own rule picks stream-K for this launch (half of the last round of workgroups would be empty)
    gstride = 2 * ((27 * 4 * (ci // 64) + 8 + 7) // 8 * 8)  # (8-byte units; the assembly loop's table entries are 16 bytes)
    assert lib0.scn_conv_workspace_bytes(27, N, N, N, ci, co, 0) == 1024 * 2 * gstride * 8 + 1024 * 128 * 64 * 4
    rules, _ = R.submanifold_rules(grid)
    rng = np.random.default_rng(ci + co)
    x = (rng.normal(size=(N, ci)) * 2 + rng.normal(size=ci)).astype(np.float32)
    w = (rng.normal(size=(27, ci, co)) * (2.0 / ci / 27) ** 0.5).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    res = rng.normal(size=(N, co)).astype(np.float32)
    dy = rng.normal(size=(N, co)).astype(np.float32)
    gamma, beta = rng.normal(size=ci).astype(np.float32), rng.normal(size=ci).astype(np.float32)
    xt, wt, bt, rt, dyt, gt, bet = [_t(a, gpu) for a in (x, w, b, res, dy, gamma, beta)]
    e = torch.empty(0, device=gpu)
    # y = BN(x): the convolution's input, so that its dInput launch can take the batch norm's backward sums
    y, sm, si = torch.empty(0, device=gpu), torch.empty(0, device=gpu), torch.empty(0, device=gpu)
    SCN.BatchNormalization_updateOutput(xt, y, sm, si, torch.zeros(ci, device=gpu), torch.ones(ci, device=gpu), gt, bet, 1e-4, 0.9, True, leak)

    def run():
        out, sink = torch.empty(0, device=gpu), []
        SCN.SubmanifoldConvolution_updateOutput(size, F3, m, y, out, wt, bt, 1, residual=rt, stats_out=sink)
        link = SCN.BnLink(xt, sm, si, gt, bet, leak)
        d_in = torch.empty(0, device=gpu)
        SCN.SubmanifoldConvolution_backward(size, F3, m, y, d_in, dyt, wt, torch.zeros(27, ci, co, device=gpu), e, 1, _bn_link=link)
        part = link.take(d_in)
        assert part is not None and int(SCN._sk_state(gpu).view(torch.int32).abs().sum().item()) == 0
        return out, sink[0], d_in, part

    before, before_fwd = SCN.STREAM_K, SCN.STREAM_K_FWD_ONLY
    SCN.STREAM_K = True  # (hand tile_passes and the control block to every launch ...
    SCN.STREAM_K_FWD_ONLY = False  # ... the dInput launch too: the default keeps stream-K to the forward direction)
    got = {}
    try:
        got["product"] = run()
        assert 1 in m.getSubmanifoldRuleBook(size, F3).passes  # (built at first use; the dInput launch walks the same lists)
        lib = _lib.use_tuning_build()
        lib.scn_set_tuning(18, 0)
        got["stream-K, every launch"] = run()
        lib.scn_set_tuning(22, 0)
        got["stream-K, items and slices drawn from counters"] = run()
        lib.scn_set_tuning(22, 1)
        lib.scn_set_tuning(16, 0)
        got["one workgroup per tile"] = run()
        lib.scn_set_tuning(0, 2)
        got["k_conv_tile2"] = run()
        lib.scn_set_tuning(0, 4)
        lib.scn_set_tuning(16, 1)
        lib.scn_set_tuning(18, 30)
    finally:
        _lib.use_product_build()
        SCN.STREAM_K, SCN.STREAM_K_FWD_ONLY = before, before_fwd
    ref = got["one workgroup per tile"]
    for name, g in got.items():
        for i, what in enumerate(("forward", "column statistics", "dInput", "batch-norm-backward sums")):
            if name == "k_conv_tile2" and i in (1, 3):
                continue  # (its statistics come per 128-row tile as well, but are not the subject here)
            assert torch.equal(g[i], ref[i]), (name, what)
    yn = y.cpu().numpy()
    want, _ = A.conv_forward(yn, w, rules, N, b, dtype=np.float64)
    assert_close(ref[0].cpu().numpy(), want + res, "stream-K forward %d->%d" % (ci, co))
    d_in_o, _, _ = A.conv_backward(yn, dy, w, rules, dtype=np.float64)
    assert_close(ref[2].cpu().numpy(), d_in_o, "stream-K dInput")
    part = ref[1].double().sum(0).cpu().numpy()
    assert_close(part[0], (want + res).sum(0), "column sums", scale=np.abs(want + res).sum(0).max())


def test_bn_statistics_folded_and_finalized_by_one_launch(gpu, monkeypatch):
    """round 5: the per-tile statistics a convolution hands to the batch norm next to it (> 256 partials) are folded and finalized
    by ONE launch (k_bn_fold_finalize: a ticket in the stream's control block, the last workgroup finalizes) instead of two:
    forward (mean, invstd, running statistics, y) and backward (dGamma, dBeta, dX) bitwise equal to the two-launch form, the
    control block zero afterwards."""
    from occuseg_amd import SCN
    ci, co = 64, 64
    m, size, grid, _, _ = _setup(gpu, seed=11, n=72000, batch=2, extent=60)
    N = grid.n
    assert (N + 127) // 128 > 256
    rng = np.random.default_rng(3)
    x = _t(rng.normal(size=(N, ci)), gpu)
    w = _t(rng.normal(size=(27, ci, co)) * 0.05, gpu)
    dy = _t(rng.normal(size=(N, co)), gpu)
    gamma, beta = _t(rng.normal(size=co), gpu), _t(rng.normal(size=co), gpu)
    e = torch.empty(0, device=gpu)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(SCN, "BN_FOLD_FUSED", fused)
        out, sink = torch.empty(0, device=gpu), []
        SCN.SubmanifoldConvolution_updateOutput(size, F3, m, x, out, w, e, 1, stats_out=sink)
        assert sink[0].size(0) > 256
        y, sm, si = torch.empty(0, device=gpu), torch.empty(0, device=gpu), torch.empty(0, device=gpu)
        rm, rv = torch.zeros(co, device=gpu), torch.ones(co, device=gpu)
        SCN.BatchNormalization_updateOutput(out, y, sm, si, rm, rv, gamma, beta, 1e-4, 0.9, True, 0.2, _colstats=[(co, sink[0])])
        # backward: the batch norm sat in front of a convolution whose dInput launch took the two reductions
        link = SCN.BnLink(out, sm, si, gamma, beta, 0.2)
        d_y = torch.empty(0, device=gpu)
        SCN.SubmanifoldConvolution_backward(size, F3, m, y, d_y, dy, w, torch.zeros(27, co, co, device=gpu), e, 1, _bn_link=link)
        part = link.take(d_y)
        assert part is not None and part.size(0) > 256
        dg, db, dx = torch.zeros(co, device=gpu), torch.zeros(co, device=gpu), torch.empty(0, device=gpu)
        SCN.BatchNormalization_backward(out, dx, y, d_y, sm, si, rm, rv, gamma, beta, dg, db, 0.2, _rewrite_d_output=False,
                                        _mask_from_x=True, _partials=part)
        assert int(SCN._sk_state(gpu).view(torch.int32).abs().sum().item()) == 0
        res[fused] = (y, sm, si, rm, rv, dg, db, dx)
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


def test_stream_k_give_up_is_a_hard_error(gpu):
    """round-4 advisor: a stream-K piece that stops waiting for its seed slab (a bounded wait) computes from an unpublished slab and
    leaves the control block poisoned; nothing on the training path looked at the word that says so.  Now every batch's read-back
    carries it: the next InputLayer call raises, behind a reset of the control blocks."""
    from occuseg_amd import SCN
    m, size, grid, _, coords = _setup(gpu, seed=2, n=3000)
    m.getSubmanifoldRuleBook(size, F3)  # (a recorded request: the next InputLayer call on this size builds it up front and reads back)
    blk = SCN._sk_state(gpu).view(torch.int32)
    assert not SCN.stream_k_gave_up()
    blk[9] = 1    # what k_conv_tile4 leaves when a wait gives up ...
    blk[40] = 1   # ... and the stale "slab published" flag of the late producer
    torch.cuda.synchronize()
    assert SCN.stream_k_gave_up()
    m2 = SCN.Metadata_3()
    c = torch.from_numpy(np.ascontiguousarray(coords))
    with pytest.raises(RuntimeError, match="stopped waiting"):
        m2.inputLayer(size, c, 0, 4, gpu)
    assert int(blk.abs().sum().item()) == 0 and not SCN.stream_k_gave_up()
    m2 = SCN.Metadata_3()
    m2.inputLayer(size, c, 0, 4, gpu)  # the next batch is fine
    assert m2.getNActive(size) == grid.n


# (C_i, C_o, points per sample, extent, strided): offset splits 3-, 9- and 27-fold (submanifold) and 8-fold (strided rulebook)
@pytest.mark.parametrize("ci,co,n,extent,strided", [(128, 128, 9000, 30, False), (512, 128, 9000, 30, False),
                                                    (64, 192, 1200, 14, False), (128, 64, 14000, 36, True)])
def test_tile4_split_launch_adds_its_partial_tiles_itself(gpu, ci, co, n, extent, strided):
    """Round 5: the coarse UNet levels (< 512 tiles: the offsets of a tile split over 3 / 9 / 27 / 8 workgroups) run on
    k_conv_tile4, whose last workgroup per (tile, column block) adds the partial tiles in slab order and emits the column
    statistics (SPLIT).  Forward with bias, shortcut and statistics and the dInput launch with the batch-norm-backward sums:
    against the float64 oracle (arithmetic of sparseconvnet/SCN/CPU/Convolution.cpp:114-189) and BITWISE against k_conv_tile2's
    split launch (same slabs, same order); the control block is zero again after every launch."""
    from occuseg_amd import SCN, _lib
    from oracle import arith as A, rules as R
    m, size, grid, _, _ = _setup(gpu, seed=31, n=n, batch=2, extent=extent)
    lib0 = _lib.load()
    e = torch.empty(0, device=gpu)
    rng = np.random.default_rng(ci + co + n)
    if strided:
        cgrid, rules, _ = R.strided_rules(grid)
        n_in, n_out, K = grid.n, cgrid.n, 8
        fwd_plan = lib0.scn_conv_plan(8, grid.n, n_in, n_out, ci, co, 0)
    else:
        rules, _ = R.submanifold_rules(grid)
        n_in = n_out = grid.n
        K = 27
        fwd_plan = lib0.scn_conv_plan(27, grid.n, n_in, n_out, ci, co, 0)
    assert fwd_plan // 100 == 4 and fwd_plan % 100 > 1, fwd_plan
    x = (rng.normal(size=(n_in, ci)) * 2 + rng.normal(size=ci)).astype(np.float32)
    w = (rng.normal(size=(K, ci, co)) * (2.0 / ci / K) ** 0.5).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    res = rng.normal(size=(n_out, co)).astype(np.float32)
    dy = rng.normal(size=(n_out, co)).astype(np.float32)
    gamma, beta = rng.normal(size=ci).astype(np.float32), rng.normal(size=ci).astype(np.float32)
    xt, wt, bt, rt, dyt, gt, bet = [_t(a, gpu) for a in (x, w, b, res, dy, gamma, beta)]
    y, sm, si = torch.empty(0, device=gpu), torch.empty(0, device=gpu), torch.empty(0, device=gpu)
    SCN.BatchNormalization_updateOutput(xt, y, sm, si, torch.zeros(ci, device=gpu), torch.ones(ci, device=gpu), gt, bet, 1e-4, 0.9, True, 0.2)

    def run():
        out, sink = torch.empty(0, device=gpu), []
        link = SCN.BnLink(xt, sm, si, gt, bet, 0.2)
        d_in = torch.empty(0, device=gpu)
        if strided:
            SCN.Convolution_updateOutput(size, size // 2, F2, F2, m, y, out, wt, bt, stats_out=sink)
            SCN.Convolution_backward(size, size // 2, F2, F2, m, y, d_in, dyt, wt, torch.zeros(K, ci, co, device=gpu), e, _bn_link=link)
        else:
            SCN.SubmanifoldConvolution_updateOutput(size, F3, m, y, out, wt, bt, 1, residual=rt, stats_out=sink)
            SCN.SubmanifoldConvolution_backward(size, F3, m, y, d_in, dyt, wt, torch.zeros(K, ci, co, device=gpu), e, 1, _bn_link=link)
        part = link.take(d_in)
        assert int(SCN._sk_state(gpu).view(torch.int32).abs().sum().item()) == 0
        return out, (sink[0] if sink else None), d_in, part

    got = {"product": run()}
    splits = set()
    try:
        lib = _lib.use_tuning_build()
        got["k_conv_tile4, split"] = run()
        lib.scn_set_tuning(7, 0)
        assert lib.scn_conv_plan(K, grid.n, n_in, n_out, ci, co, 0) == 200 + fwd_plan % 100
        got["k_conv_tile2, split"] = run()
        # other splits of the same launch (the workgroups a split launch aims for: tuning key 4), each bitwise against k_conv_tile2
        for policy in (600, 9000):
            lib.scn_set_tuning(4, policy)
            lib.scn_set_tuning(7, 1)
            code = lib.scn_conv_plan(K, grid.n, n_in, n_out, ci, co, 0)
            assert code // 100 == 4
            splits.add(code % 100)
            a = run()
            lib.scn_set_tuning(7, 0)
            b_ = run()
            assert torch.equal(a[0], b_[0]) and torch.equal(a[2], b_[2]), ("policy", policy, code)
            got["k_conv_tile4, plan %d" % code] = a
        lib.scn_set_tuning(4, 1500)
        lib.scn_set_tuning(7, 1)
    finally:
        _lib.use_product_build()
    splits.add(fwd_plan % 100)
    assert splits == ({8} if strided else {27} if n_out <= 2048 else {3, 9, 27}), splits
    ref = got["k_conv_tile2, split"]
    for name, g in got.items():
        if "plan" not in name:  # (another split is another summation order: against the oracle below)
            assert torch.equal(g[0], ref[0]), (name, "forward")
            assert torch.equal(g[2], ref[2]), (name, "dInput")
        assert (g[1] is None) == (n_out <= 2048), (name, "column statistics")
    yn = y.cpu().numpy()
    want, _ = A.conv_forward(yn, w, rules, n_out, b, dtype=np.float64)
    if not strided:
        want = want + res
    d_in_o, _, _ = A.conv_backward(yn, dy, w, rules, dtype=np.float64)
    if not strided:  # (the dInput of a strided convolution writes every row once: k_conv_expand, not a split launch)
        assert lib0.scn_conv_plan(27, grid.n, n_out, n_in, co, ci, 0) // 100 == 4
    for name, g in got.items():
        assert_close(g[0].cpu().numpy(), want, "%s: forward %d->%d" % (name, ci, co))
        assert_close(g[2].cpu().numpy(), d_in_o, "%s: dInput" % name)
    out, stats, d_in, part = got["product"]
    if stats is not None:
        s = stats.double().sum(0).cpu().numpy()
        assert_close(s[0], want.sum(0), "column sums", scale=np.abs(want).sum(0).max())
        assert_close(s[1], (want * want).sum(0), "column sums of squares")
    if part is not None and not strided:
        # the two reductions of the batch norm's backward pass over the oracle's dInput, with the device's mask
        mask = np.where(yn > 0, 1.0, 0.2)
        dprime = d_in_o * mask
        p = part.double().sum(0).cpu().numpy()
        mean = sm.double().cpu().numpy()
        assert_close(p[0], dprime.sum(0), "sum d'", scale=np.abs(dprime).sum(0).max())
        assert_close(p[1], ((x - mean) * dprime).sum(0), "sum (x - mean) d'", scale=np.abs((x - mean) * dprime).sum(0).max())


def test_tile4_equals_tile2_bitwise_sweep(gpu):
    """300 randomised launches (clouds of varying density / extent / batch, C_i 64..384, C_o 64..256, submanifold and
    strided rulebooks in both directions, plain / transposed / reversed weights, with and without bias) through
    k_conv_tile4 and through k_conv_tile2: same summation order, so every output must be BIT-identical -- any
    scheduling hazard in the hand-written load pipeline of k_conv_tile4 shows up as a difference"""
    import os
    import sys
    from occuseg_amd import SCN, _lib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_variants
    try:
        ran4, bad = stress_variants.run(300, 0, max_points=150000, verbose=False)
    finally:
        _lib.use_product_build()
        SCN.invalidate_weight_cache()
    assert bad == 0 and ran4 >= 100, (ran4, bad)


@pytest.mark.parametrize("ci,co,n", [(64, 64, 72000), (32, 32, 3000), (128, 64, 3000), (3, 16, 3000)])
def test_submanifold_forward_with_fused_residual(gpu, ci, co, n):
    """out = residual + bias + conv(x) inside the kernel (the shortcut of a residual block): every forward kernel
    (barrier-free tile, LDS-staged tile with and without the offset split, generic) against conv + residual"""
    from occuseg_amd import SCN
    from oracle import arith as A, rules as R
    m, size, grid, _, _ = _setup(gpu, seed=5, n=n, batch=2, extent=60 if n > 10000 else 20)
    rules, _ = R.submanifold_rules(grid)
    rng = np.random.default_rng(ci * co)
    x = rng.normal(size=(grid.n, ci)).astype(np.float32)
    w = (rng.normal(size=(27, ci, co)) * (2.0 / ci / 27) ** 0.5).astype(np.float32)
    b = rng.normal(size=co).astype(np.float32)
    res = rng.normal(size=(grid.n, co)).astype(np.float32)
    want, _ = A.conv_forward(x, w, rules, grid.n, b, dtype=np.float64)
    out = torch.empty(0, device=gpu)
    SCN.SubmanifoldConvolution_updateOutput(size, F3, m, _t(x, gpu), out, _t(w, gpu), _t(b, gpu), 1, residual=_t(res, gpu))
    assert_close(out.cpu().numpy(), want + res, "conv + residual %d->%d" % (ci, co))


def test_residual_block_fusion_equals_add_table(gpu):
    """scn.UNet's residual blocks hand the shortcut to their last convolution (ResidualConcatTable); a plain
    ConcatTable -> AddTable over the SAME modules gives the same outputs and the same gradients"""
    import occuseg_amd
    from occuseg_amd import scenes
    scn = occuseg_amd.install()
    torch.manual_seed(9)
    m = 16
    net = scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
        scn.UNet(3, 1, [m, 2 * m], residual_blocks=True)).add(scn.OutputLayer(3)).to(gpu)
    fused = [mod for mod in net.modules() if isinstance(mod, scn.ResidualConcatTable)]
    assert len(fused) == 3
    coords, feats = scenes.batch([1000, 1001], target_sites=4000)
    batch = [torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 2]

    def run():
        for p in net.parameters():
            p.grad = None
        out = net(batch)
        out.square().sum().backward()
        return out.detach().clone(), [p.grad.clone() for p in net.parameters()]

    o1, g1 = run()
    for mod in fused:
        mod.__class__ = scn.ConcatTable   # same children, plain behaviour: [shortcut, body] -> AddTable
    o2, g2 = run()
    for mod in fused:
        mod.__class__ = scn.ResidualConcatTable
    assert_close(o1.cpu().numpy(), o2.cpu().numpy(), "fused residual output", rel=1e-5)
    for a, b in zip(g1, g2):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), "fused residual grads", rel=1e-4)


@pytest.mark.parametrize("case", ["tile4", "tile2", "split", "expand"])
def test_conv_column_statistics_for_batchnorm(gpu, case):
    """scn_conv_forward's colstats: per block of output rows the column sums / sums of squares of the RESULT, taken from
    the output tile while it is in LDS (or registers) -- every emitting kernel against a torch reduction of the output,
    and the batch norm finalized from them against the ordinary two-pass batch norm."""
    from occuseg_amd import SCN
    from oracle import rules as R
    big = case in ("tile4", "tile2")  # >= 512 output tiles: no offset split (tile2: C_i = 32 is not tile4's)
    m, size, grid, _, _ = _setup(gpu, seed=21, n=72000 if big else (900 if case == "split" else 9000), batch=2,
                                 extent=60 if big else 26)
    ci, co = {"tile4": (64, 64), "tile2": (32, 64), "split": (128, 128), "expand": (128, 64)}[case]
    rng = np.random.default_rng(3)
    stats = []
    out = torch.empty(0, device=gpu)
    if case == "expand":  # deconvolution: coarse -> fine, one contribution per fine row
        rbk = m.getRuleBook(size, size // 2, F2, F2)
        n_coarse = m.getNActive(size // 2)
        x = torch.randn(n_coarse, ci, device=gpu)
        w = torch.randn(8, ci, co, device=gpu) * 0.1
        b = torch.randn(co, device=gpu)
        SCN.Deconvolution_updateOutput(size // 2, size, F2, F2, m, x, out, w, b, stats_out=stats)
        assert rbk.unique_col == 0
    else:
        x = torch.randn(grid.n, ci, device=gpu)
        w = torch.randn(27, ci, co, device=gpu) * 0.05
        b = torch.randn(co, device=gpu)
        res = torch.randn(grid.n, co, device=gpu)
        SCN.SubmanifoldConvolution_updateOutput(size, F3, m, x, out, w, b, 1, residual=res, stats_out=stats)
    if case == "split":  # launches that split their offsets (coarse levels) emit none: the batch norm reads its input itself
        assert stats == [] and SCN._lib_().scn_conv_workspace_bytes(27, grid.n, grid.n, grid.n, ci, co, 0) > 0
        return
    assert len(stats) == 1 and stats[0].shape[1:] == (2, co)
    cs = stats[0].double().sum(0).cpu().numpy()
    o64 = out.double()
    assert_close(cs[0], o64.sum(0).cpu().numpy(), case + " column sums", rel=1e-5, scale=float(o64.abs().sum(0).max()))
    assert_close(cs[1], (o64 * o64).sum(0).cpu().numpy(), case + " column sums of squares", rel=1e-5)
    # batch norm from the handed-over statistics == batch norm with its own statistics pass
    g, beta = torch.rand(co, device=gpu) + 0.5, torch.randn(co, device=gpu)
    ys = []
    for ext in (None, [(co, stats[0])]):
        y, sm, si = torch.empty(0, device=gpu), torch.empty(0, device=gpu), torch.empty(0, device=gpu)
        rm, rv = torch.zeros(co, device=gpu), torch.ones(co, device=gpu)
        SCN.BatchNormalization_updateOutput(out, y, sm, si, rm, rv, g, beta, 1e-4, 0.9, True, 0.0, _colstats=ext)
        ys.append((y, sm, si, rm, rv))
    for a, bb, what in zip(ys[0], ys[1], ("y", "mean", "invstd", "running_mean", "running_var")):
        assert_close(bb.cpu().numpy(), a.cpu().numpy(), "bn from colstats: " + what, rel=2e-5)
    # two producers (a concatenation): segments in order
    if case == "tile2":
        cat = torch.cat([out, out], 1).contiguous()
        y2 = torch.empty(0, device=gpu)
        SCN.BatchNormalization_updateOutput(cat, y2, torch.empty(0, device=gpu), torch.empty(0, device=gpu),
                                            torch.zeros(2 * co, device=gpu), torch.ones(2 * co, device=gpu),
                                            torch.cat([g, g]), torch.cat([beta, beta]), 1e-4, 0.9, True, 0.0,
                                            _colstats=[(co, stats[0]), (co, stats[0])])
        assert torch.equal(y2[:, :co], ys[1][0]) and torch.equal(y2[:, co:], ys[1][0])


def test_network_with_and_without_statistics_handover(gpu):
    """the UNet with the convolutions handing their column statistics to the batch norms (default) against the same
    network with every batch norm reading its input twice: same outputs / gradients to rounding, and both bitwise
    reproducible"""
    import occuseg_amd
    from occuseg_amd import scenes
    from occuseg_amd.sparseconvnet import conv as conv_mod
    scn = occuseg_amd.install()
    torch.manual_seed(4)
    mm = 64
    net = scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(scn.SubmanifoldConvolution(3, 3, mm, 3, False)).add(
        scn.UNet(3, 1, [mm, 2 * mm], residual_blocks=True, leakiness=1)).add(scn.BatchNormalization(mm)).add(
        scn.OutputLayer(3)).to(gpu)  # leakiness 1: no activation mask that a last-bit difference of a statistic could flip
    coords, feats = scenes.batch([1000, 1001], target_sites=40000)  # level 0: >= 512 tiles (no split), level 1: split
    batch = [torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 2]
    bns = [mod for mod in net.modules() if isinstance(mod, scn.BatchNormalization)]
    used = []
    hooks = [b.register_forward_pre_hook(lambda mod, inp: used.append(inp[0].get_colstats() is not None)) for b in bns]

    def run():
        for p in net.parameters():
            p.grad = None
        for b in bns:
            b.running_mean.zero_()
            b.running_var.fill_(1)
        out = net(batch)
        out.square().sum().backward()
        return out.detach().clone(), [p.grad.clone() for p in net.parameters()]

    o1, g1 = run()
    n_ext = sum(used)
    # level 0: every batch norm but the first (the stem runs on the generic kernel) takes the statistics of its producer(s),
    # the concatenation [encoder features | deconvolution] from two of them; the split launches of level 1 emit none
    assert 4 <= n_ext < len(bns), (n_ext, len(bns))
    o1b, _ = run()
    assert torch.equal(o1, o1b)
    try:
        conv_mod.EMIT_COLSTATS = False
        del used[:]
        o2, g2 = run()
        assert not any(used)
    finally:
        conv_mod.EMIT_COLSTATS = True
        for h in hooks:
            h.remove()
    assert_close(o1.cpu().numpy(), o2.cpu().numpy(), "statistics hand-over: output", rel=2e-5)
    for a, b in zip(g1, g2):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), "statistics hand-over: grads", rel=2e-4)


# dInput launches: 64->64 and 128->64 on k_conv_tile4, 96->64 on the unsplit k_conv_tile2; the fourth (a small grid) is an
# offset-split launch whose last workgroup per tile adds the partial tiles and emits the sums (round 4); the fifth (<= 2048
# rows) emits none: the batch norm then runs its own one-launch pass
@pytest.mark.parametrize("ci,co,n,extent,leak", [(64, 64, 72000, 60, 0.0), (64, 96, 72000, 60, 0.2), (128, 64, 20000, 40, 1.0),
                                                 (64, 64, 3000, 20, 0.0), (64, 64, 900, 12, 0.0)])
def test_bn_backward_reductions_from_the_dinput_launch(gpu, ci, co, n, extent, leak):
    """scn_conv_forward(bn_bwd): the dInput launch of a convolution whose input was y = BN(x) leaves sum d' and
    sum (x - mean) d' per block of rows; scn_bn_backward_ext finishes the batch norm's backward pass from them, with another
    gradient added inside the apply pass -- against the oracle's batch-norm backward (arithmetic of
    sparseconvnet/SCN/CPU/BatchNormalization.cpp:64-107) over the oracle's dInput, and against the plain two-pass path."""
    from occuseg_amd import SCN
    from oracle import arith as A, rules as R
    m, size, grid, _, _ = _setup(gpu, seed=3, n=n, batch=2, extent=extent)
    rules, _ = R.submanifold_rules(grid)
    rng = np.random.default_rng(ci + co)
    N = grid.n
    x = (rng.normal(size=(N, ci)) * 2 + rng.normal(size=ci)).astype(np.float32)
    gamma, beta = rng.normal(size=ci).astype(np.float32), rng.normal(size=ci).astype(np.float32)
    w = (rng.normal(size=(27, ci, co)) * (2.0 / ci / 27) ** 0.5).astype(np.float32)
    dy = rng.normal(size=(N, co)).astype(np.float32)
    extra = rng.normal(size=(N, ci)).astype(np.float32)
    e = torch.empty(0, device=gpu)
    xt, gt, bt = _t(x, gpu), _t(gamma, gpu), _t(beta, gpu)
    y, sm, si = torch.empty(0, device=gpu), torch.empty(0, device=gpu), torch.empty(0, device=gpu)
    rm, rv = torch.zeros(ci, device=gpu), torch.ones(ci, device=gpu)
    SCN.BatchNormalization_updateOutput(xt, y, sm, si, rm, rv, gt, bt, 1e-4, 0.9, True, leak)
    # the convolution's backward pass with the link: d_in = d(loss)/dy, partial sums for the batch norm
    link = SCN.BnLink(xt, sm, si, gt, bt, leak)
    d_in, gw = torch.empty(0, device=gpu), torch.zeros(27, ci, co, device=gpu)
    SCN.SubmanifoldConvolution_backward(size, F3, m, y, d_in, _t(dy, gpu), _t(w, gpu), gw, e, 1, _bn_link=link)
    partials = link.take(d_in)
    from occuseg_amd import _lib
    code = _lib.load().scn_conv_plan(27, N, N, N, co, ci, 0)
    assert (n > 3000) == (code % 100 == 1) and (n == 900) == (N <= 2048)
    assert (partials is None) == (N <= 2048)
    assert partials is None or partials.shape[1:] == (2, ci)
    dgam, dbet = torch.zeros(ci, device=gpu), torch.zeros(ci, device=gpu)
    dx = torch.empty(0, device=gpu)
    SCN.BatchNormalization_backward(xt, dx, y, d_in, sm, si, rm, rv, gt, bt, dgam, dbet, leak, _rewrite_d_output=False,
                                    _mask_from_x=True, _partials=partials, _d_x_add=_t(extra, gpu))
    # plain path on the same d_in
    dgam2, dbet2, dx2 = torch.zeros(ci, device=gpu), torch.zeros(ci, device=gpu), torch.empty(0, device=gpu)
    SCN.BatchNormalization_backward(xt, dx2, y, d_in, sm, si, rm, rv, gt, bt, dgam2, dbet2, leak, _rewrite_d_output=False,
                                    _mask_from_x=True)
    assert_close(dx.cpu().numpy(), dx2.cpu().numpy() + extra, "fused dX against the two-pass path", rel=1e-5)
    assert_close(dgam.cpu().numpy(), dgam2.cpu().numpy(), "dGamma", rel=1e-5)
    assert_close(dbet.cpu().numpy(), dbet2.cpu().numpy(), "dBeta", rel=1e-5)
    # oracle: batch-norm backward of the oracle's own dInput, with the device's y for the mask
    yo, smo, sio, _, _ = A.batchnorm_forward(x, gamma, beta, np.zeros(ci), np.ones(ci), train=True, leakiness=leak, dtype=np.float64)
    d_in_o, _, _ = A.conv_backward(y.cpu().numpy(), dy, w, rules, dtype=np.float64)
    want_dx, _, want_dg, want_db = A.batchnorm_backward(x, y.cpu().numpy(), d_in_o, smo, sio, gamma, leakiness=leak, dtype=np.float64)
    assert_close(dx.cpu().numpy(), want_dx + extra, "fused dX against the oracle")
    assert_close(dgam.cpu().numpy(), want_dg, "dGamma against the oracle")
    assert_close(dbet.cpu().numpy(), want_db, "dBeta against the oracle")
    # a gradient that was modified after the launch does not get the stale sums
    link2 = SCN.BnLink(xt, sm, si, gt, bt, leak)
    d2 = torch.empty(0, device=gpu)
    SCN.SubmanifoldConvolution_backward(size, F3, m, y, d2, _t(dy, gpu), _t(w, gpu), torch.zeros(27, ci, co, device=gpu), e, 1,
                                        _bn_link=link2)
    d2.add_(1.0)
    assert link2.take(d2) is None


@pytest.mark.parametrize("p,n,extent", [(64, 72000, 60), (128, 20000, 40), (64, 3000, 20)])
def test_deconvolution_into_and_out_of_a_joined_tensor(gpu, p, n, extent):
    """scn_conv_forward's ld_out / ld_in, scn_conv_dweight's ld_dout, scn_bn_backward_ext's ld_add: a deconvolution writes its
    result into the RIGHT half of a [rows, 2p] tensor (k_conv_expand), its backward pass reads the right half of a joined
    gradient in place (dInput through k_conv_tile4 / k_conv_tile2 incl. the offset split, dWeight through k_dw_tile), a batch
    norm's apply pass adds the LEFT half -- all against the same calls on contiguous copies (bitwise for the forward and the
    dInput, whose summation order does not change; 1e-5 for dWeight: float atomics) and the forward against the oracle."""
    from occuseg_amd import SCN
    from oracle import arith as A, rules as R
    m, size, grid, _, _ = _setup(gpu, seed=17, n=n, batch=2, extent=extent)
    coarse, srules, _ = R.strided_rules(grid)
    half = size // 2
    m.getRuleBook(size, half, F2, F2)
    nf, nc = grid.n, coarse.n
    ci, co = 2 * p, p  # deconvolution 2p -> p (coarse -> fine)
    rng = np.random.default_rng(p)
    xc = rng.normal(size=(nc, ci)).astype(np.float32)
    w = (rng.normal(size=(8, ci, co)) * (2.0 / ci / 8) ** 0.5).astype(np.float32)
    e = torch.empty(0, device=gpu)
    # forward into the right half of a joined buffer
    plain = torch.empty(0, device=gpu)
    SCN.Deconvolution_updateOutput(half, size, F2, F2, m, _t(xc, gpu), plain, _t(w, gpu), e)
    want, _ = A.conv_forward(xc, w, srules, nf, None, in_col=1, out_col=0, dtype=np.float64)
    assert_close(plain.cpu().numpy(), want, "deconvolution forward")
    if SCN.deconvolution_can_write_a_slice(half, size, F2, F2, m, ci, co):
        joined = torch.full((nf, 2 * co), 7.0, device=gpu)
        SCN.Deconvolution_updateOutput(half, size, F2, F2, m, _t(xc, gpu), torch.empty(0, device=gpu), _t(w, gpu), e,
                                       _out_view=joined[:, co:])
        assert torch.equal(joined[:, co:], plain) and bool((joined[:, :co] == 7.0).all())
    else:
        assert n == 3000  # planes % 64 == 0 everywhere else
    # backward out of the right half of a joined gradient
    g = torch.from_numpy(rng.normal(size=(nf, 2 * co)).astype(np.float32)).to(gpu)
    g_right = g[:, co:]
    assert not g_right.is_contiguous()
    res = {}
    for name, d_out in (("slice", g_right), ("copy", g_right.contiguous())):
        d_in, d_w = torch.empty(0, device=gpu), torch.zeros(8, ci, co, device=gpu)
        SCN.Deconvolution_backward(half, size, F2, F2, m, _t(xc, gpu), d_in, d_out, _t(w, gpu), d_w, e)
        res[name] = (d_in, d_w)
    assert torch.equal(res["slice"][0], res["copy"][0])
    assert_close(res["slice"][1].cpu().numpy(), res["copy"][1].cpu().numpy(), "dWeight from a column slice", rel=1e-5)
    d_in_o, d_w_o, _ = A.conv_backward(xc, g_right.cpu().numpy(), w, srules, in_col=1, out_col=0, dtype=np.float64)
    assert_close(res["slice"][0].cpu().numpy(), d_in_o, "deconvolution dInput from a column slice")
    assert_close(res["slice"][1].cpu().numpy(), d_w_o, "deconvolution dWeight from a column slice")
    # batch norm backward adding the LEFT half of the joined gradient inside its apply pass
    x = torch.from_numpy(rng.normal(size=(nf, co)).astype(np.float32)).to(gpu)
    gam, bet = torch.from_numpy(rng.normal(size=co).astype(np.float32)).to(gpu), torch.zeros(co, device=gpu)
    y, sm, si = torch.empty(0, device=gpu), torch.empty(0, device=gpu), torch.empty(0, device=gpu)
    SCN.BatchNormalization_updateOutput(x, y, sm, si, torch.zeros(co, device=gpu), torch.ones(co, device=gpu), gam, bet,
                                        1e-4, 0.9, True, 0.0)
    dy = torch.from_numpy(rng.normal(size=(nf, co)).astype(np.float32)).to(gpu)
    outs = []
    for add in (g[:, :co], g[:, :co].contiguous()):
        dx = torch.empty(0, device=gpu)
        SCN.BatchNormalization_backward(x, dx, y, dy, sm, si, None, None, gam, bet, torch.zeros(co, device=gpu),
                                        torch.zeros(co, device=gpu), 0.0, _rewrite_d_output=False, _mask_from_x=True,
                                        _d_x_add=add)
        outs.append(dx)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("ci,co,n,extent", [(64, 64, 200000, 110), (128, 64, 60000, 70), (96, 32, 9000, 34), (64, 128, 700, 12)])
def test_dweight_fixed_order_inside_the_kernel(gpu, ci, co, n, extent):
    """scn_conv_dweight with a workspace AND dw_state: every workgroup's block of partial sums goes to its own slot, the last
    workgroup of each group of 16 slices adds the group, the last group adds the groups and updates dW -- no float atomics, no
    memset, no second launch.  Against the float64 oracle; bitwise equal run to run and with a delay that changes which
    workgroup arrives last; the two-launch form (no dw_state) and the atomic form agree at 1e-5; the ticket block is zero
    again after every launch; a dW that is not zero is added to.  (~190 k sites: ~90 slices per list = 6 groups, two levels;
    ~57 k sites: 2 groups; 9,000 and 700 sites: one group, no second level; 96 -> 32: the 32-plane blocks.)"""
    from occuseg_amd import SCN, _lib
    from occuseg_amd.SCN import ptr, stream, check
    from oracle import arith as A, rules as R
    m, size, grid, _, _ = _setup(gpu, seed=31, n=n, batch=2, extent=extent)
    rb = m.getSubmanifoldRuleBook(size, F3)
    rules, _ = R.submanifold_rules(grid)
    N = grid.n
    rng = np.random.default_rng(ci + co)
    x = rng.normal(size=(N, ci)).astype(np.float32)
    dy = rng.normal(size=(N, co)).astype(np.float32)
    w = np.zeros((27, ci, co), np.float32)
    _, want, _ = A.conv_backward(x, dy, w, rules, dtype=np.float64)
    lib = _lib.load()
    xt, dyt = _t(x, gpu), _t(dy, gpu)
    wsb = lib.scn_conv_dweight_workspace_bytes(27, rb.stride, N, N, ci, co, 1)
    state = torch.zeros(_lib.DW_STATE_BYTES, dtype=torch.uint8, device=gpu)

    def run(mode, dw0=None, poison=False):
        dw = torch.zeros(27, ci, co, device=gpu) if dw0 is None else dw0.clone()
        ws = torch.empty(wsb, dtype=torch.uint8, device=gpu) if mode else None
        if poison and ws is not None:
            ws.fill_(0x7f)  # (the fused form never reads a slot it has not written: NaN-ish garbage must not show)
        check(lib.scn_conv_dweight(ptr(xt), ptr(dyt), ptr(dw), ptr(rb.rules), ptr(rb.counts), ptr(rb.bases.get(1)), 27,
                                   rb.stride, 0, 1, N, N, ci, co, 0, 0, ptr(ws), wsb if mode else 0,
                                   ptr(state) if mode == 2 else None, stream()))
        return dw

    fused = run(2, poison=True)
    assert int(state.view(torch.int32).abs().sum().item()) == 0
    assert_close(fused.cpu().numpy(), want, "dW, fixed order inside the kernel", rel=1e-4)
    for _ in range(3):
        assert torch.equal(run(2, poison=True), fused)
        assert int(state.view(torch.int32).abs().sum().item()) == 0
    two = run(1)
    atomic = run(0)
    assert_close(two.cpu().numpy(), fused.cpu().numpy(), "two-launch against in-kernel", rel=1e-5)
    assert_close(atomic.cpu().numpy(), fused.cpu().numpy(), "atomic against in-kernel", rel=1e-5)
    base = torch.randn(27, ci, co, device=gpu)
    assert_close(run(2, dw0=base).cpu().numpy(), (base + fused).cpu().numpy(), "dW += ...", rel=1e-6)


def test_dweight_fixed_order_one_list_many_slices(gpu):
    """NetworkInNetwork's dWeight is ONE list (the identity rulebook) cut into up to 1024 slices: groups of 32.  600,000 rows,
    128 -> 64: 938 slices; against x^T dy in float64, bitwise equal run to run, tickets zero afterwards."""
    from occuseg_amd import SCN, _lib
    from occuseg_amd.SCN import ptr, stream, check
    n, ci, co = 600000, 128, 64
    torch.manual_seed(3)
    x, dy = torch.randn(n, ci, device=gpu), torch.randn(n, co, device=gpu)
    rb = SCN._identity_rulebook(n, gpu)
    lib = _lib.load()
    wsb = lib.scn_conv_dweight_workspace_bytes(1, rb.stride, n, n, ci, co, int(rb.bases.get(1) is not None))
    assert wsb // (ci * co * 4) > 256  # (more slices than two levels of 16 cover)
    state = torch.zeros(_lib.DW_STATE_BYTES, dtype=torch.uint8, device=gpu)
    outs = []
    for _ in range(3):
        dw = torch.zeros(ci, co, device=gpu)
        ws = torch.empty(wsb, dtype=torch.uint8, device=gpu).fill_(0x7f)
        check(lib.scn_conv_dweight(ptr(x), ptr(dy), ptr(dw), ptr(rb.rules), ptr(rb.counts), ptr(rb.bases.get(1)), 1, rb.stride,
                                   0, 1, n, n, ci, co, 0, 0, ptr(ws), wsb, ptr(state), stream()))
        assert int(state.view(torch.int32).abs().sum().item()) == 0
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = (x.double().t() @ dy.double()).cpu().numpy()
    assert_close(outs[0].cpu().numpy(), want, "NiN dW, fixed order inside the kernel", rel=1e-4)
