"""Test helper: the module tree of the reference's ``LearningBWDenseUNet`` /
``InstanceDenseUNet`` (examples/ScanNet/model.py:657-717) rebuilt from the mirror's operator API -- same attribute
names (hence the same state_dict keys), same construction ORDER (hence the same parameters for the same torch seed:
tests/golden/make_g6.py checks ``torch.equal`` against the reference class itself, built from the reference file).
The reference file cannot travel to the GPU box; this restatement of its ~40 structural lines can."""
import torch
import torch.nn as nn


class InstanceDenseUNet(nn.Module):
    def __init__(self, scn, config):
        nn.Module.__init__(self)
        self.config = config
        self.input = scn.InputLayer(config["dimension"], config["full_scale"], mode=4,
                                    normal_guide_scale=(config["full_scale"] >> config["rotation_guide_level"]) + 1)
        self.sub = scn.SubmanifoldConvolution(config["dimension"], config["input_feature_number"],
                                              config["unet_structure"][0], 3, False)
        self.unet = scn.UNet(config["dimension"], config["block_reps"], config["unet_structure"],
                             config["residual_blocks"])
        d = self.output_feature_dim = config["unet_structure"][0]
        self.bn = scn.BatchNormReLU(d)
        self.output = scn.OutputLayer(d)
        self.linear = nn.Linear(d, config["class_num"])
        self.fc_regress = nn.Linear(d, d)
        self.linear_regress = nn.Linear(d, 1)
        self.sigmoid_regress = nn.Sigmoid()
        self.fc_embedding = nn.Linear(d, d)
        self.linear_embedding = nn.Linear(d, d)
        self.fc_displacement = nn.Linear(d, d)
        self.linear_displacement = nn.Linear(d, config["dimension"])

    def forward(self, x):
        feature = self.output(self.bn(self.unet(self.sub(self.input(x)))))
        y = self.linear(feature)
        embedding = self.linear_embedding(self.fc_embedding(feature))
        offset = self.sigmoid_regress(self.linear_regress(self.fc_regress(feature)))
        displacement = self.linear_displacement(self.fc_displacement(feature))
        return y, feature, embedding, offset, displacement


class LearningBWDenseUNet(nn.Module):
    def __init__(self, scn, config):
        nn.Module.__init__(self)
        self.config = config
        self.backbone = InstanceDenseUNet(scn, config)
        d = self.backbone.output_feature_dim
        self.fc_bw = nn.Linear(d, d)
        self.linear_bw = nn.Linear(d, 2)
        self.relu_bw = nn.Softplus()
        self.fc_occupancy = nn.Linear(d, d)
        self.linear_occupancy = nn.Linear(d, 1)
        self.relu_occupancy = nn.Softplus()

    def forward(self, x):
        semantics, feature, embedding, offset, displacement = self.backbone(x)
        bw = self.relu_bw(self.linear_bw(self.fc_bw(feature)))
        occupancy = self.relu_occupancy(self.linear_occupancy(self.fc_occupancy(feature)))
        return semantics, feature, embedding, offset, displacement, bw, occupancy


OUTPUT_NAMES = ("semantics", "feature", "embedding", "offset", "displacement", "bw", "occupancy")


def config(m=16):
    return {"dimension": 3, "full_scale": 4096, "rotation_guide_level": 0, "input_feature_number": 3,
            "unet_structure": [m, 2 * m, 3 * m, 4 * m, 5 * m, 6 * m], "block_reps": 1, "residual_blocks": True,
            "class_num": 20}


def oracle_forward(net, coords, feats, dtype=torch.float64):
    """the same forward in float64 on the CPU: backbone through oracle/torch_ref.py (oracle rulebooks, index_select ->
    mm -> index_add_), heads through torch itself.  Returns the 7 outputs as numpy arrays."""
    import occuseg_amd
    from oracle import torch_ref
    scn = occuseg_amd.install()
    bb = net.backbone
    seq = scn.Sequential().add(bb.input).add(bb.sub).add(bb.unet).add(bb.bn).add(bb.output)  # the net's own modules
    feature, _ = torch_ref.ref_forward(seq, coords, feats, dtype)
    feature = feature.detach()

    def lin(mod, x):
        return x @ mod.weight.detach().cpu().to(dtype).t() + mod.bias.detach().cpu().to(dtype)

    y = lin(bb.linear, feature)
    embedding = lin(bb.linear_embedding, lin(bb.fc_embedding, feature))
    offset = torch.sigmoid(lin(bb.linear_regress, lin(bb.fc_regress, feature)))
    displacement = lin(bb.linear_displacement, lin(bb.fc_displacement, feature))
    bw = torch.nn.functional.softplus(lin(net.linear_bw, lin(net.fc_bw, feature)))
    occupancy = torch.nn.functional.softplus(lin(net.linear_occupancy, lin(net.fc_occupancy, feature)))
    return [t.numpy() for t in (y, feature, embedding, offset, displacement, bw, occupancy)]
