"""-m gpu: SURVEY.md section 8 f1 -- the reference's ``LearningBWDenseUNet`` (backbone + 9 nn.Linear heads,
examples/ScanNet/model.py:657-717) on the GPU backend against tests/golden/g6_instance_dense_unet.npz, a fixture
generated in the build container by tests/golden/make_g6.py from the reference's own file (float64 oracle).  The module
tree is rebuilt here by tests/instance_unet.py with the same seed; make_g6.py asserted that this rebuild has
bit-identical parameters to the reference class."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_instance_dense_unet.npz")


def test_learning_bw_dense_unet_matches_reference_fixture(gpu):
    import instance_unet as IU
    import occuseg_amd
    scn = occuseg_amd.install()
    g = np.load(FIXTURE)
    torch.manual_seed(int(g["seed"]))
    net = IU.LearningBWDenseUNet(scn, IU.config(int(g["m"])))
    checksum = sum(float(v.double().sum()) for v in net.state_dict().values())
    assert abs(checksum - float(g["checksum_params"])) <= 1e-9 * abs(checksum), "parameter initialisation drifted"
    net = net.to(gpu).train()
    # the reference's input convention: float coordinates [P,4] (x,y,z,sample), features, zeros(3,3) = no normals, batch
    x = [torch.from_numpy(g["locs"]).to(gpu), torch.from_numpy(g["feats"]).to(gpu), torch.zeros(3, 3), 2]
    outs = net(x)
    assert len(outs) == 7
    for name, got in zip(IU.OUTPUT_NAMES, outs):
        assert_close(got.detach().cpu().numpy(), g[name], name)
    # backward through the heads and the backbone runs and reaches every parameter
    sum(o.sum() for o in outs).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    # host int64 coordinates (what sparseconvnet/ioLayers.py:56 turns them into) give the same sites
    x2 = [torch.from_numpy(g["locs"]).long(), x[1], torch.zeros(3, 3), 2]
    net.zero_grad()
    outs2 = net(x2)
    assert torch.equal(outs2[1], outs[1])
