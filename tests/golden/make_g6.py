#!/usr/bin/env python3
"""Generate tests/golden/g6_instance_dense_unet.npz.  BUILD CONTAINER ONLY: needs /root/reference.

Imports the reference's examples/ScanNet/model.py UNCHANGED (its `import sparseconvnet as scn` resolves to the mirror),
seeds torch, builds ``LearningBWDenseUNet`` at m = 16 from THAT file, checks that tests/instance_unet.py's rebuild of
the module tree has bit-identical parameters for the same seed, evaluates the reference-class network with the float64
oracle (oracle/torch_ref.py + torch heads) on a 2-scene batch in the reference's input convention
``[locs float [P,4], feats, zeros(3,3), batch]`` and stores the inputs and the 7 outputs (< 1 MB, float32).
The -m gpu test (tests/test_gpu_instance_unet.py) rebuilds the tree from the mirror with the same seed and compares
the device's 7 outputs with these at 1e-4."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
MODEL = "/root/reference/examples/ScanNet/model.py"
SEED, M, SITES = 1234, 16, 1500


def load_reference_model():
    import occuseg_amd
    from occuseg_amd import segment
    scn = occuseg_amd.install()
    # application-side imports of model.py: tensorboardX is not used by the model classes; torch_scatter's functions
    # are provided by occuseg_amd.segment (scatter_add/mean/std/max), the rest are never called by these classes
    tb = types.ModuleType("tensorboardX")
    tb.SummaryWriter = None
    sys.modules.setdefault("tensorboardX", tb)
    ts = types.ModuleType("torch_scatter")
    for name in ("scatter_max", "scatter_mean", "scatter_std", "scatter_add"):
        setattr(ts, name, getattr(segment, name))
    for name in ("scatter_sub", "scatter_min", "scatter_div"):
        setattr(ts, name, None)
    sys.modules.setdefault("torch_scatter", ts)
    spec = importlib.util.spec_from_file_location("occuseg_reference_model", MODEL)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return scn, ref


def main():
    import instance_unet as IU
    from occuseg_amd import scenes
    scn, ref = load_reference_model()
    cfg = IU.config(M)
    torch.manual_seed(SEED)
    net_ref = ref.LearningBWDenseUNet(cfg)           # the reference's class, from the reference's file
    torch.manual_seed(SEED)
    net_mirror = IU.LearningBWDenseUNet(scn, cfg)    # the test helper's rebuild
    sd_a, sd_b = net_ref.state_dict(), net_mirror.state_dict()
    assert list(sd_a) == list(sd_b), "state_dict keys differ"
    assert all(torch.equal(sd_a[k], sd_b[k]) for k in sd_a), "same seed must give the same parameters"
    coords, feats = scenes.batch([2000, 2001], target_sites=SITES)
    locs = coords.astype(np.float32) + np.random.default_rng(SEED).uniform(0, 0.99, (len(coords), 4)).astype(np.float32)
    locs[:, 3] = coords[:, 3]                        # float coordinates as trainMerge produces them; truncation = voxel
    assert np.array_equal(locs.astype(np.int64), coords)
    net_ref.train()
    outs = IU.oracle_forward(net_ref, locs.astype(np.int64), feats)
    data = {"locs": locs, "feats": feats, "seed": np.int64(SEED), "m": np.int64(M),
            "checksum_params": np.float64(sum(float(v.double().sum()) for v in sd_a.values()))}
    for name, o in zip(IU.OUTPUT_NAMES, outs):
        data[name] = o.astype(np.float32)
    out = os.path.join(ROOT, "tests", "golden", "g6_instance_dense_unet.npz")
    np.savez_compressed(out, **data)
    print("wrote", out, os.path.getsize(out), "bytes;", {k: v.shape for k, v in data.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
