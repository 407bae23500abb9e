"""-m gpu: a train step is a train step.  The convolutions cache their weights re-laid out in MFMA fragment order
(occuseg_amd/SCN.py: weight epochs); the reference reads the module's weight tensor on every call
(sparseconvnet/SCN/CUDA/Convolution.cpp:105-162).  Round 2 keyed that cache on the tensor's version counter, which
torch's fused optimizers and writes through `.data` do not move: forward and dInput kept running on first-step weights
and no test took an optimizer step.  These do."""
import numpy as np
import pytest
import torch

from helpers import assert_close, device_masks

pytestmark = pytest.mark.gpu


def _net(scn, m, levels):
    planes = [m * (i + 1) for i in range(levels)]
    return scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(
        scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
        scn.UNet(3, 1, planes, residual_blocks=True)).add(
        scn.BatchNormReLU(m)).add(scn.OutputLayer(3))


def _check_against_oracle(net, batch, coords, feats, what):
    from occuseg_amd import SCN
    from oracle import torch_ref
    with torch.no_grad():
        got = net(batch)
        before = SCN.WPACK_CACHE
        SCN.WPACK_CACHE = False  # every call re-packs from the module's weights: the reference behaviour, literally
        try:
            uncached = net(batch)
        finally:
            SCN.WPACK_CACHE = before
    want, _ = torch_ref.ref_forward(net, coords, feats, torch.float64)  # reads the parameters as they are NOW
    assert_close(got.cpu().numpy(), want.detach().numpy(), what + ": forward on the updated parameters")
    assert torch.equal(got, uncached), what + ": cached packed weights differ from freshly packed ones"


@pytest.mark.parametrize("fused", [True, False])
def test_forward_and_dinput_follow_the_optimizer(gpu, fused):
    """three Adam steps (fused: the version counter does not move), then a write through .data"""
    import occuseg_amd
    from occuseg_amd import SCN, scenes
    scn = occuseg_amd.install()
    torch.manual_seed(11)
    coords, feats = scenes.batch([1000, 1001], target_sites=5000)
    net = _net(scn, 16, 3).to(gpu)
    batch = [torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 2]
    opt = torch.optim.Adam(net.parameters(), lr=3e-2, fused=fused)  # a step that moves every weight by ~3e-2
    first = net[1].weight.detach().clone()
    for step in range(3):
        opt.zero_grad()
        net(batch).sum().backward()
        opt.step()
        _check_against_oracle(net, batch, coords, feats, "after Adam step %d (fused=%s)" % (step, fused))
    assert (net[1].weight.detach() - first).abs().max().item() > 1e-2  # the weights really moved
    # steady state: ONE multi-tensor re-pack per forward pass, no per-layer launches
    SCN.WPACK_STATS.update(single=0, batch=0, batch_entries=0)
    opt.zero_grad()
    net(batch).sum().backward()
    opt.step()
    assert SCN.WPACK_STATS["single"] == 0 and SCN.WPACK_STATS["batch"] == 1, SCN.WPACK_STATS
    # dInput after the update: the whole backward pass against the float64 oracle on the updated parameters
    from oracle import torch_ref
    opt.zero_grad()
    out, masks = device_masks(net, scn, lambda: net(batch))
    gsel = torch.from_numpy(np.random.default_rng(0).normal(size=tuple(out.shape)))
    (out * gsel.to(gpu, torch.float32)).sum().backward()
    # (the oracle differentiates over the device's own activation masks: every gradient at 1e-4, not the 1e-2 of round 3)
    ref_out, stats = torch_ref.ref_forward(net, coords, feats, torch.float64, masks=masks)
    (ref_out * gsel).sum().backward()
    for mod in net.modules():
        for name, leaf in (stats["params"].get(id(mod)) or {}).items():
            assert_close(getattr(mod, name).grad.cpu().numpy(), leaf.grad.numpy(), "grad %s.%s" % (mod, name), rel=1e-4)
    # a write through .data (no version bump, no optimizer): the next forward pass starts with an InputLayer call
    with torch.no_grad():
        for p in net.parameters():
            p.data.add_(0.05 * torch.randn_like(p))
    _check_against_oracle(net, batch, coords, feats, "after p.data.add_()")


def test_reused_input_tensor_and_manual_sgd(gpu):
    """a loop that calls the InputLayer ONCE and updates the weights by hand through .data (no optimizer object, no
    version bump): the first forward-direction call after a backward call starts a new weight epoch"""
    import occuseg_amd
    from occuseg_amd import SCN, scenes
    scn = occuseg_amd.install()
    torch.manual_seed(12)
    coords, feats = scenes.batch([1002], target_sites=4000)
    net = _net(scn, 16, 2).to(gpu)
    x0 = net[0]([torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 1])
    body = list(net.children())[1:]

    def run(x):
        for mod in body:
            x = mod(x)
        return x

    for step in range(3):
        for p in net.parameters():
            p.grad = None
        run(x0).sum().backward()
        for p in net.parameters():
            p.data.add_(-3e-2 * torch.sign(p.grad))
        with torch.no_grad():
            got = run(x0)
            SCN.WPACK_CACHE = False
            try:
                want = run(x0)
            finally:
                SCN.WPACK_CACHE = True
        assert torch.equal(got, want), step
    # and without any signal at all -- a write through `.data` between two forward sweeps over the SAME built input, under
    # no_grad, no optimizer, no backward, no InputLayer call (rounds 3-5: "call SCN.invalidate_weight_cache() yourself") -- the
    # next sweep sees the new weights by construction: the outermost container of a sweep starts a weight epoch, a convolution
    # called outside any container re-packs its own weights (SCN.py "Forward scopes"; the reference reads the module's
    # weights on every call, sparseconvnet/SCN/CUDA/Convolution.cpp:105-162)
    with torch.no_grad():
        before = run(x0)
        for p in net.parameters():
            p.data.mul_(1.5)
        after = run(x0)
        SCN.WPACK_CACHE = False
        try:
            want = run(x0)
        finally:
            SCN.WPACK_CACHE = True
    assert torch.equal(after, want) and not torch.equal(before, after)


def test_colstats_follow_the_features_assignment(gpu):
    """the column statistics a convolution hands to the next batch norm belong to ONE assignment of `features`"""
    import occuseg_amd
    from occuseg_amd import scenes
    scn = occuseg_amd.install()
    torch.manual_seed(13)
    coords, feats = scenes.batch([1003], target_sites=70000)  # >= 512 tiles: a launch that emits the statistics
    inp = scn.InputLayer(3, 4096, mode=4)
    stem = scn.SubmanifoldConvolution(3, 3, 64, 3, False).to(gpu)
    conv_ = scn.SubmanifoldConvolution(3, 64, 64, 3, False).to(gpu)
    bn = scn.BatchNormReLU(64).to(gpu)

    def conv(t):
        return conv_(stem(t))
    x = conv(inp([torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 1]))
    assert x.get_colstats() is not None
    bn(x)  # consumes the handed-over statistics
    x.features = x.features * 2.0 + 1.0  # re-assigned: the statistics are gone, the batch norm reads the tensor
    assert x.get_colstats() is None
    bn2 = scn.BatchNormReLU(64).to(gpu)
    y1 = bn2(x).features
    f = x.features.detach().double()
    want = torch.relu((f - f.mean(0)) / torch.sqrt(f.var(0, unbiased=False) + bn2.eps))
    assert_close(y1.detach().cpu().numpy(), want.cpu().numpy(), "batch norm of the re-assigned features")
    x2 = conv(inp([torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 1]))
    with torch.no_grad():
        x2.features.mul_(3.0)  # in place: version counter
    assert x2.get_colstats() is None


def test_deterministic_mode_train_steps_are_bitwise_equal(gpu):
    """The DEFAULT (SCN.DETERMINISTIC_DW, on since round 4): every dWeight launch adds its workgroups' partial sums in a fixed
    order -- inside the kernel for k_dw_tile (slots, arrival tickets, two levels), memset + kernel + reduction for the stem and
    the narrow fallbacks -- instead of with float atomics.  Forward, dInput and batch norm are deterministic anyway, so two runs of three whole
    train steps (fused Adam included) from the same seed end in bitwise-equal parameters -- which the reference's own
    backward pass cannot do (float atomics, sparseconvnet/SCN/CUDA/Convolution.cu:749) -- and the gradients equal the default
    mode's at 1e-5.  m64 levels 0-1 at 2 x 40k sites (k_dw_tile over 64-plane blocks, the stem's k_dw_rows, NiN) and
    m16 (the 32-plane and generic fallbacks)."""
    import occuseg_amd
    from occuseg_amd import SCN, dp, scenes
    scn = occuseg_amd.install()

    assert SCN.DETERMINISTIC_DW is True  # (no switch: the default)

    def run(m, levels, sites, deterministic, steps=3):
        if not deterministic:
            SCN.DETERMINISTIC_DW = False
        torch.manual_seed(77)
        net = _net(scn, m, levels).to(gpu)
        coords, feats = scenes.batch([1000, 1001], target_sites=sites)
        batch = [torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 2]
        bucket = dp.FlatGradBucket(net.parameters())
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
        first = None
        try:
            for _ in range(steps):
                bucket.zero()
                net(batch).sum().backward()
                bucket.all_reduce_mean()
                if first is None:
                    first = bucket.flat.clone()
                opt.step()
        finally:
            bucket.close()
            SCN.DETERMINISTIC_DW = True
        return first, [p.detach().clone() for p in net.parameters()]

    try:
        for m, levels, sites in ((64, 2, 40000), (16, 3, 6000)):
            g1, p1 = run(m, levels, sites, True)
            g2, p2 = run(m, levels, sites, True)
            assert torch.equal(g1, g2), "first-step gradients differ between two deterministic runs (m%d)" % m
            assert all(torch.equal(a, b) for a, b in zip(p1, p2)), "parameters after 3 steps differ (m%d)" % m
            g0, _ = run(m, levels, sites, False, steps=1)
            assert_close(g1.cpu().numpy(), g0.cpu().numpy(), "deterministic against atomic gradients", rel=1e-5)
    finally:
        SCN.DETERMINISTIC_DW = True


def test_poisoned_ticket_word_is_a_hard_error(gpu):
    """The tickets of the split launches, of the batch norm's fold and of the deterministic dWeight share the stream's control
    blocks with stream-K's flags and rely on every launch leaving them ZERO; a launch that dies after drawing tickets would make
    every later one sum incomplete partial results, silently (round-5 advisor).  FlatGradBucket.zero() looks at the blocks at
    every step boundary (one step late, without waiting for the device): a poisoned word raises and the blocks are reset."""
    import occuseg_amd
    from occuseg_amd import SCN, dp, scenes
    scn = occuseg_amd.install()
    torch.manual_seed(3)
    coords, feats = scenes.batch([1000], target_sites=6000)
    net = _net(scn, 16, 3).to(gpu)
    batch = [torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 1]
    bucket = dp.FlatGradBucket(net.parameters())
    try:
        for _ in range(3):  # healthy steps: nothing raised, tickets back at zero
            bucket.zero()
            net(batch).sum().backward()
            bucket.all_reduce_mean()
        block = SCN._sk_state(gpu)
        torch.cuda.synchronize()
        assert int(block.view(torch.int32).abs().sum().item()) == 0
        block.view(torch.int32)[40] = 1  # a ticket word that some launch "left behind"
        bucket.zero()                    # queues the check that sees it ...
        with pytest.raises(RuntimeError, match="control block"):
            bucket.zero()                # ... and the next step boundary raises
        torch.cuda.synchronize()
        assert int(block.view(torch.int32).abs().sum().item()) == 0  # reset
        bucket.zero()
        net(batch).sum().backward()      # and training goes on
        bucket.all_reduce_mean()
        bucket.zero()
        bucket.zero()
    finally:
        bucket.close()
        SCN._BLOCK_CHECKS.clear()
