"""not gpu, build container only: SURVEY.md section 8(f1) -- the reference's own model definition
(examples/ScanNet/model.py, imported from /root/reference WITHOUT modification) constructs on top of
the mirror, with the parameter layout the README's baseline_m64.pth would load into.  The forward
pass of exactly this architecture runs on the GPU in tests/test_gpu_unet.py / bench.py (the reference
tree does not exist on the GPU box, so the file itself can only be imported here)."""
import importlib.util
import os
import sys
import types

import pytest
import torch

MODEL = "/root/reference/examples/ScanNet/model.py"


@pytest.mark.skipif(not os.path.exists(MODEL), reason="reference tree not present")
def test_instance_dense_unet_builds_unchanged():
    import occuseg_amd
    scn = occuseg_amd.install()
    # application-side dependencies that are absent in this image and irrelevant to the backbone
    for name, attrs in (("tensorboardX", ["SummaryWriter"]),
                        ("torch_scatter", ["scatter_max", "scatter_mean", "scatter_std", "scatter_sub", "scatter_min",
                                           "scatter_add", "scatter_div"])):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            for a in attrs:
                setattr(mod, a, None)
            sys.modules[name] = mod
    spec = importlib.util.spec_from_file_location("occuseg_reference_model", MODEL)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)  # `import sparseconvnet as scn` inside resolves to the mirror
    m = 64
    config = {"dimension": 3, "full_scale": 4096, "rotation_guide_level": 0, "input_feature_number": 3,
              "unet_structure": [m, 2 * m, 3 * m, 4 * m, 5 * m, 6 * m], "block_reps": 1, "residual_blocks": True,
              "class_num": 20}
    net = ref.LearningBWDenseUNet(config)
    bb = net.backbone
    assert isinstance(bb.input, scn.InputLayer) and bb.input.mode == 4
    assert isinstance(bb.unet, scn.Sequential)
    n_backbone = sum(p.numel() for mod in (bb.sub, bb.unet, bb.bn) for p in mod.parameters())
    assert n_backbone == 43434432
    sd = net.state_dict()
    assert sd["backbone.sub.weight"].shape == (27, 3, 64)
    assert sd["backbone.unet.0.1.1.weight"].shape == (27, 64, 64)
    assert sd["backbone.unet.2.1.1.weight"].shape == (8, 64, 128)
    assert sd["backbone.bn.running_var"].shape == (64,)
    # a checkpoint saved from the reference has exactly these keys: round-trip through state_dict
    net2 = ref.LearningBWDenseUNet(config)
    net2.load_state_dict(sd, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))


@pytest.mark.skipif(not os.path.exists(MODEL), reason="reference tree not present")
def test_fixture_g6_belongs_to_the_reference_class():
    """tests/golden/g6_instance_dense_unet.npz was generated from the reference's own model.py (make_g6.py): the
    reference class, seeded the same way, still has the parameters the fixture recorded, and tests/instance_unet.py's
    rebuild (what the GPU test uses) is bit-identical to it."""
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    sys.path.insert(0, here)
    import instance_unet as IU
    import make_g6
    scn, ref = make_g6.load_reference_model()
    g = np.load(os.path.join(here, "golden", "g6_instance_dense_unet.npz"))
    torch.manual_seed(int(g["seed"]))
    a = ref.LearningBWDenseUNet(IU.config(int(g["m"])))
    torch.manual_seed(int(g["seed"]))
    b = IU.LearningBWDenseUNet(scn, IU.config(int(g["m"])))
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    checksum = sum(float(v.double().sum()) for v in sa.values())
    assert abs(checksum - float(g["checksum_params"])) <= 1e-9 * abs(checksum)
