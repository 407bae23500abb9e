"""not gpu: the C-ABI library loads and exports every symbol include/scn_hip.h declares (no compute
calls), and the host-side mirror keeps the reference's operator interface."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "scn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from occuseg_amd import _lib
    names = _declared()
    assert len(names) >= 25
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    lib = _lib.load()  # raises if the .so is missing or a symbol cannot be resolved
    for n in names:
        assert hasattr(lib, n), n
    assert lib.scn_abi_version() == 7
    # tuning / ablation entry points are not part of the product library (only of libscn_hip_tuning.so)
    assert not hasattr(lib, "scn_set_tuning") and not hasattr(lib, "scn_debug_stamps")
    assert os.path.exists(_lib.TUNING_LIB_PATH)
    # big launches (k_conv_tile4) keep one pass table per workgroup in the workspace (116 entries -> 120: whole 64-byte lines,
    # x 16 bytes since round 6: the assembly pass loop's entries carry the weight offset and the lane masks); as stream-K launches -- 4688 tiles are 4.58 rounds of 1024 resident workgroups, 1383 tiles x 2 column blocks 2.70:
    # both would leave more than 3 % of the last round empty -- two pass tables per persistent workgroup + one 128 x 64 fp32 slab
    # per slice; 4096 tiles are whole rounds: one workgroup per tile.  A coarse level (few tiles) splits its offsets 9- or 27-fold
    assert lib.scn_conv_workspace_bytes(27, 600000, 600000, 600000, 64, 64, 0) == 1024 * 2 * 120 * 16 + 1024 * 128 * 64 * 4
    assert lib.scn_conv_workspace_bytes(27, 177001, 177001, 177001, 128, 128, 0) == 1024 * 2 * 224 * 16 + 1024 * 128 * 64 * 4
    assert lib.scn_conv_workspace_bytes(27, 524288, 524288, 524288, 64, 64, 0) == 4096 * 120 * 16
    assert lib.scn_conv_workspace_bytes(27, 600000, 600000, 600000, 20, 12, 0) == 0  # generic kernel: none
    # (round 5: on k_conv_tile4 too -- 16 tiles x 6 column blocks x 27 workgroups: the 27 slabs, then a pass table of one offset x six
    #  chunks x <= 4 passes + 8 sentinels = 32 entries per workgroup)
    assert lib.scn_conv_plan(27, 2000, 2000, 2000, 384, 384, 0) == 427
    assert lib.scn_conv_workspace_bytes(27, 2000, 2000, 2000, 384, 384, 0) == 27 * 2000 * 384 * 4 + 16 * 6 * 27 * 32 * 16
    assert lib.scn_conv_plan(27, 7000, 7000, 7000, 256, 256, 0) == 409 and lib.scn_conv_stats_blocks(27, 7000, 7000, 7000, 256, 256, 0) == 55
    assert lib.scn_conv_stats_blocks(27, 2000, 2000, 2000, 384, 384, 0) == 0  # (<= 2048 rows: the batch norm's one-launch form)
    # size queries are host-only arithmetic: safe without a GPU
    assert lib.scn_hash_capacity(1000) == 2048
    assert lib.scn_conv_wpack_bytes(27, 64, 64) == 27 * 64 * 64 * 4
    assert lib.scn_bn_temp_bytes(64) > 2 * 64 * 4
    assert lib.scn_rules_temp_bytes(1000, 27) >= 27 * 4 * 4


def test_occupancy_box_plan_is_host_arithmetic():
    """scn_occ_plan (round 6): the bounding box of a batch as bitmaps -- origin aligned to the coarsest scale (scale l = scale 0
    shifted by l), extents halved per scale, one 64-bit word per 32 voxels of an x row, scales one behind the other; a box beyond
    SCN_OCC_MAX_WORDS is refused by size 0, not by an error; an occupancy layout of a batch has neither hash maps nor a sort's
    workspace"""
    import ctypes
    from occuseg_amd import _lib
    lib = _lib.load()
    mm = (ctypes.c_int32 * 8)(37, 1001, 515, 2, 126, 1090, 604, 4)
    box = _lib.OccBox()
    assert lib.scn_occ_plan(mm, 4, ctypes.byref(box)) == 0
    assert list(box.origin) == [32, 1000, 512, 2] and box.nb == 3 and box.levels == 4
    W = 0
    for l in range(4):
        n = [((mm[4 + a] - box.origin[a]) >> l) + 1 for a in range(3)]
        assert list(box.n[l]) == n and box.wx[l] == (n[0] + 31) // 32 and box.woff[l] == W
        W += 3 * n[2] * n[1] * box.wx[l]
    assert box.woff[4] == W and box.blocks == (W + 1023) // 1024
    assert box.off_info == 4096 and box.off_blk == 5120 and box.off_words % 256 == 0 and box.off_words >= box.off_blk + 4 * (box.blocks + 1)
    assert box.bytes == box.off_words + box.blocks * 1024 * 8
    big = (ctypes.c_int32 * 8)(0, 0, 0, 0, 1023, 2047, 1023, 7)
    assert lib.scn_occ_plan(big, 6, ctypes.byref(box)) == 0 and box.bytes == 0
    assert lib.scn_occ_plan(mm, 0, ctypes.byref(box)) != 0 and lib.scn_occ_plan(mm, 13, ctypes.byref(box)) != 0
    bad = (ctypes.c_int32 * 8)(5, 5, 5, 0, 4, 9, 9, 0)
    assert lib.scn_occ_plan(bad, 2, ctypes.byref(box)) != 0
    plan = _lib.BatchPlan()
    plan.levels, plan.coords_dtype, plan.ncols, plan.P, plan.flags = 2, 0, 4, 50000, 1
    for l, n in enumerate((30000, 9000)):
        plan.N[l], plan.sub[l], plan.strided[l] = n, 1, 1 if l == 0 else 0
    lo = _lib.BatchLayout()
    assert lib.scn_batch_layout_query(ctypes.byref(plan), ctypes.byref(lo)) == 0
    assert all(lo.hash_cap[l] == 0 and lo.hash_keys[l] == -1 and lo.hash_vals[l] == -1 for l in range(2))
    # scratch: arrival numbers [P], per-site counts, block sums; the larger of the two rulebook scratches behind them
    assert lo.total - lo.scratch >= 4 * 50000 + 4 * 30001 + 27 * 30000 * 4


def test_product_path_has_no_cpu_fallback_and_never_imports_the_oracle():
    from occuseg_amd import SCN
    x = torch.zeros(4, 16)
    e = torch.empty(0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        SCN.BatchNormalization_updateOutput(x, e, e, e, torch.zeros(16), torch.ones(16), e, e, 1e-4, 0.9, True, 1.0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        SCN.SubmanifoldConvolution_updateOutput(torch.tensor([8, 8, 8]), torch.tensor([3, 3, 3]), SCN.Metadata_3(),
                                                x, e, torch.zeros(27, 16, 16), e, 1)
    pkg = os.path.join(ROOT, "occuseg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_out_of_scope_entry_points_raise():
    from occuseg_amd import SCN
    import occuseg_amd.sparseconvnet as scn
    with pytest.raises(NotImplementedError):
        SCN.MaxPooling_updateOutput()
    with pytest.raises(NotImplementedError):
        scn.MaxPooling
    m = SCN.Metadata_3()
    with pytest.raises(NotImplementedError):
        m.getSubmanifoldRuleBook(torch.tensor([8, 8, 8]), torch.tensor([5, 5, 5]))
    with pytest.raises(NotImplementedError):
        m.getRuleBook(torch.tensor([8, 8, 8]), torch.tensor([3, 3, 3]), torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2]))


def test_unet_module_tree_matches_reference_layout():
    """state_dict keys / shapes of InstanceDenseUNet's backbone: 43,434,432 parameters
    at m=64, weight [volume, nIn, nOut], BN buffers running_mean/running_var."""
    import occuseg_amd
    scn = occuseg_amd.install()
    import sparseconvnet
    assert sparseconvnet is scn
    m = 64
    net = scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
        scn.UNet(3, 1, [m, 2 * m, 3 * m, 4 * m, 5 * m, 6 * m], residual_blocks=True)).add(scn.BatchNormReLU(m)).add(
        scn.OutputLayer(3))
    assert sum(p.numel() for p in net.parameters()) == 43434432
    sd = net.state_dict()
    assert sd["1.weight"].shape == (27, 3, 64)
    assert sd["2.0.1.1.weight"].shape == (27, 64, 64)            # first residual block, first SC
    assert sd["2.2.1.1.weight"].shape == (8, 64, 128)            # level-0 strided convolution
    assert sd["2.2.1.4.weight"].shape == (8, 128, 64)            # matching deconvolution
    assert sd["2.4.0.weight"].shape == (128, 64)                 # NiN shortcut of the decoder block
    assert sd["2.4.1.1.weight"].shape == (27, 128, 64)
    assert "2.0.1.0.running_mean" in sd and "2.0.1.0.running_var" in sd and "3.bias" in sd
    kinds = [type(mod).__name__ for mod in net.modules()]
    assert kinds.count("SubmanifoldConvolution") == 23 and kinds.count("Convolution") == 5
    assert kinds.count("Deconvolution") == 5 and kinds.count("NetworkInNetwork") == 5
    assert sum(k.startswith("BatchNorm") for k in kinds) == 33
    bn = scn.BatchNormalization(8)
    assert (bn.eps, bn.momentum, bn.leakiness) == (1e-4, 0.9, 1)
    assert scn.BatchNormReLU(8).leakiness == 0 and scn.BatchNormLeakyReLU(8).leakiness == 0.333


def test_spatial_size_arithmetic():
    import occuseg_amd.sparseconvnet as scn
    c = scn.Convolution(3, 4, 8, 2, 2, False)
    d = scn.Deconvolution(3, 8, 4, 2, 2, False)
    s = torch.tensor([4096, 4096, 4096])
    assert c.output_spatial_size(s).tolist() == [2048] * 3 and c.output_spatial_size(s).dtype == torch.int64
    assert d.output_spatial_size(torch.tensor([2048] * 3)).tolist() == [4096] * 3
    assert c.input_spatial_size(torch.tensor([2048] * 3)).tolist() == [4096] * 3
    with pytest.raises(AssertionError):
        c.output_spatial_size(torch.tensor([4097, 4096, 4096]))
    assert scn.toLongTensor(3, 5).tolist() == [5, 5, 5] and scn.toLongTensor(3, [1, 2, 3]).tolist() == [1, 2, 3]
    assert scn.optionalTensor(c, "bias").numel() == 0 and scn.optionalTensorReturn(torch.empty(0)) is None
    assert repr(c) == "Convolution 4->8 C2/2" and repr(scn.SubmanifoldConvolution(3, 4, 8, 3, False)) == \
        "SubmanifoldConvolution 4->8 C3"


def test_scenes_are_deterministic_and_inside_the_key_box():
    import numpy as np
    from occuseg_amd import scenes
    a, fa = scenes.batch([1000, 1001], 4000)
    b, fb = scenes.batch([1000, 1001], 4000)
    assert np.array_equal(a, b) and np.array_equal(fa, fb)
    assert a[:, 0].max() < 1024 and a[:, 1].max() < 2048 and a[:, 2].max() < 1024 and a[:, :3].min() >= 10
    assert np.all(np.diff(a[:, 3]) >= 0) and set(a[:, 3]) == {0, 1}
    assert len(scenes.solid_block(6)) == 216 and scenes.bernoulli_grid(32, 0.25, 0).shape[1] == 3


def test_no_kernel_uses_scratch():
    """no kernel of the product library spills or indexes private arrays (tools/check_scratch.py: why it matters)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_scratch", os.path.join(ROOT, "tools", "check_scratch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ks = mod.kernels(os.path.join(ROOT, "occuseg_amd", "libscn_hip.so"))
    assert len(ks) > 100
    bad = [(n, s) for n, s, _, _ in ks if s > 0 and not mod.ALLOWED.search(n)]
    assert not bad, bad


def test_roctx_ranges_wrap_the_boundary_functions():
    """SCN.enable_roctx(): a roctx range around every boundary function (the reference's EASY_FUNCTION ranges,
    sparseconvnet/SCN/CUDA/Convolution.cpp:112-154), labelled with the function and its tensor shapes; balanced when the call
    raises; nothing at all when off."""
    from occuseg_amd import SCN

    class Fake(object):
        def __init__(self):
            self.log = []

        def roctxRangePushA(self, label):
            self.log.append(label.decode())
            return 0

        def roctxRangePop(self):
            self.log.append("pop")
            return 0

    fake = Fake()
    x = torch.zeros(5, 16)
    before = SCN._ROCTX[0]
    plain = SCN.BatchNormalization_updateOutput
    try:
        SCN._install_traced(True)
        assert SCN.BatchNormalization_updateOutput is not plain and SCN.BatchNormalization_updateOutput.untraced is plain
        SCN._ROCTX[0] = fake
        e, v = torch.empty(0), torch.zeros(16)
        with pytest.raises(RuntimeError):  # CPU tensors are refused -- inside the range
            SCN.BatchNormalization_updateOutput(x, e, e, e, v, v, v, v, 1e-4, 0.9, True, 0.0)
        assert fake.log == ["BatchNormalization_updateOutput [5x16]", "pop"]
        SCN._ROCTX[0] = None
        with pytest.raises(RuntimeError):
            SCN.BatchNormalization_updateOutput(x, e, e, e, v, v, v, v, 1e-4, 0.9, True, 0.0)
        assert len(fake.log) == 2
    finally:
        SCN._ROCTX[0] = before
        SCN._install_traced(before is not None)
    assert before is not None or SCN.BatchNormalization_updateOutput is plain  # off: not even a wrapper call


def test_size_keys_are_cached_per_tensor_and_version():
    """SCN._size_key: the modules hand the same few size tensors to every call; the tuple is cached per tensor object and
    follows an in-place change"""
    import torch
    from occuseg_amd import SCN
    t = torch.tensor([4096, 4096, 2048])
    assert SCN._size_key(t) == (4096, 4096, 2048) and SCN._size_key(t) == (4096, 4096, 2048)
    assert SCN._SIZE_KEYS[id(t)][0] is t
    t[2] = 4096
    assert SCN._size_key(t) == (4096, 4096, 4096)
    assert SCN._size_key([3, 3, 3]) == (3, 3, 3) and SCN._size_key(torch.tensor([2, 2, 2]) // 2) == (1, 1, 1)


def test_forward_scopes_start_one_weight_epoch_per_sweep():
    """SCN.py "Forward scopes": the OUTERMOST container of a forward sweep starts a new weight epoch (so that a write through
    `.data` between two sweeps over the same built input is seen, as in the reference, which reads the weights on every
    call), nested containers do not, and an exception inside a container leaves the depth where it was."""
    import occuseg_amd
    from occuseg_amd import SCN
    scn = occuseg_amd.install()
    seen = []

    class Probe(scn.Identity):
        def forward(self, input):
            seen.append((SCN._FWD_DEPTH[0], SCN._WEPOCH[0]))
            return input

    class Boom(scn.Identity):
        def forward(self, input):
            raise ValueError("boom")

    net = scn.Sequential().add(Probe()).add(scn.Sequential().add(Probe()).add(scn.ConcatTable().add(Probe()).add(Probe())))
    e0 = SCN._WEPOCH[0]
    assert SCN._FWD_DEPTH[0] == 0
    net("x")
    assert [d for d, _ in seen] == [1, 2, 3, 3] and {e for _, e in seen} == {e0 + 1} and SCN._FWD_DEPTH[0] == 0
    net("x")
    assert seen[-1][1] == e0 + 2  # the next sweep: the next epoch
    bad = scn.Sequential().add(scn.Sequential().add(Boom()))
    with pytest.raises(ValueError):
        bad("x")
    assert SCN._FWD_DEPTH[0] == 0 and SCN._WEPOCH[0] == e0 + 3
