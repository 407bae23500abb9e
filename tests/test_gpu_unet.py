"""-m gpu: the ScanNet UNet (InputLayer -> stem -> scn.UNet -> BNReLU -> OutputLayer) end to end,
forward + backward, against the float64 torch-autograd oracle on the same seeded scene."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _net(scn, m, levels, leak):
    planes = [m * (i + 1) for i in range(levels)]
    return scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(
        scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
        scn.UNet(3, 1, planes, residual_blocks=True, leakiness=leak)).add(
        scn.BatchNormReLU(m)).add(scn.OutputLayer(3))


@pytest.mark.parametrize("m,levels,leak", [(16, 3, 0.0), (32, 4, 0.2)])
def test_unet_forward_backward(gpu, m, levels, leak):
    import occuseg_amd
    from occuseg_amd import scenes
    from oracle import torch_ref
    scn = occuseg_amd.install()
    torch.manual_seed(m)
    coords, feats = scenes.batch([1000, 1001], target_sites=5000)
    net = _net(scn, m, levels, leak)
    ref_out, stats = torch_ref.ref_forward(net, coords, feats, torch.float64)
    gsel = torch.from_numpy(np.random.default_rng(0).normal(size=tuple(ref_out.shape)))
    (ref_out * gsel).sum().backward()

    net = net.to(gpu)
    scn.forward_pass_multiplyAdd_count = 0
    out = net([torch.from_numpy(coords), torch.from_numpy(feats).to(gpu), None, 2])
    (out * gsel.to(gpu, torch.float32)).sum().backward()
    assert scn.forward_pass_multiplyAdd_count > 0
    assert_close(out.detach().cpu().numpy(), ref_out.detach().numpy(), "UNet output")
    # gradients: the per-op tests (test_gpu_ops.py) hold every backward kernel to 1e-4 with the ReLU
    # mask taken from the device's own y.  End to end that cannot be done: an element with |y| ~ 1e-7
    # has a different sign in fp32 and fp64, its mask flips, and a summed gradient moves by one whole
    # term (observed: 1.2e-3 of the tensor's magnitude for one BN gamma).  Bar here: 1e-2.
    for mod in net.modules():
        ref = stats["params"].get(id(mod))
        if not ref:
            continue
        for name, leaf in ref.items():
            got = getattr(mod, name).grad
            assert got is not None, (mod, name)
            assert_close(got.cpu().numpy(), leaf.grad.numpy(), "grad %s.%s" % (mod, name), rel=1e-2)
        if id(mod) in stats["bn"]:
            mean, var = stats["bn"][id(mod)]
            assert_close(mod.running_mean.cpu().numpy(), (0.1 * mean).numpy(), "running_mean")
            assert_close(mod.running_var.cpu().numpy(), (0.9 + 0.1 * var).numpy(), "running_var")


def test_rulebook_prefetch_is_equivalent(gpu):
    """InputLayer.prefetch (grids + rulebooks built one step ahead on a side stream) must give the same
    rulebooks, outputs and gradients as building inside the step; a different coords object must not adopt."""
    import occuseg_amd
    from occuseg_amd import SCN, scenes
    scn = occuseg_amd.install()
    torch.manual_seed(3)
    net = _net(scn, 16, 3, 0.0).to(gpu)
    inp = next(mod for mod in net.modules() if isinstance(mod, scn.InputLayer))
    coords_a, feats_a = scenes.batch([1000, 1001], target_sites=6000)
    coords_b, feats_b = scenes.batch([1002, 1003], target_sites=7000)
    batches = [[torch.from_numpy(c).to(gpu), torch.from_numpy(f).to(gpu), None, 2]
               for c, f in ((coords_a, feats_a), (coords_b, feats_b))]

    def run(batch):
        for p in net.parameters():
            p.grad = None
        out = net(batch)
        out.sum().backward()
        return out.detach().clone(), [p.grad.clone() for p in net.parameters()]

    run(batches[0])  # records the rulebook requests (the plan) of this architecture
    plain = [run(b) for b in batches]
    piped = []
    inp.prefetch(batches[0])
    for i, b in enumerate(batches):
        assert id(b[0]) in SCN._PREFETCHED
        res = run(b)
        assert not SCN._PREFETCHED  # adopted
        if i + 1 < len(batches):
            inp.prefetch(batches[i + 1], wait_for_current_stream=False)
        piped.append(res)
    torch.cuda.synchronize()
    for (o0, g0), (o1, g1) in zip(plain, piped):
        # small levels split the offsets over workgroups and dWeight accumulates with float atomics: tolerance
        assert_close(o1.cpu().numpy(), o0.cpu().numpy(), "prefetch output", rel=1e-5)
        for a, b in zip(g0, g1):
            assert_close(b.cpu().numpy(), a.cpu().numpy(), "prefetch grads", rel=1e-5)
    # a prefetch for another tensor object is ignored, and dropped by the next prefetch
    inp.prefetch(batches[0])
    clone = [batches[0][0].clone()] + batches[0][1:]
    o, _ = run(clone)
    assert_close(o.cpu().numpy(), plain[0][0].cpu().numpy(), "no adoption", rel=1e-5)
    assert id(batches[0][0]) in SCN._PREFETCHED
    inp.prefetch(batches[1])
    assert list(SCN._PREFETCHED) == [id(batches[1][0])]
    SCN._PREFETCHED.clear()
