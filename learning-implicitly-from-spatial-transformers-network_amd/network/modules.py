"""nn.Module mirror of the reference's modules that LIST / CoarseNet use (network/modules.py of the
reference: PerceptualPooling :15-59, PointMLP :62-104, TreeGraphDecoder :107-132, VoxelDecoder /
VoxelDecoder2 :192-282, VoxelEncoder2 :401-442, ResEncoder :1027-1074).

Same class names, constructor arguments, forward signatures, attribute and state-dict names.  The two
hot-path modules (PerceptualPooling, VoxelDecoder2) run on the HIP library; the per-image encoders
stay PyTorch (MIOpen convolutions), as SURVEY section 8 scopes them.
"""
import torch
import torch.nn as nn

from .. import hip
from ..layers.gcn import TreeGCN
from . import hotpath
from .resnet import resnet18


# ======================================================================================= hot path
class PerceptualPooling(nn.Module):
    """Project points with the 4x3 spatial-transformer matrix and sample the five image feature
    maps, resized to map_size^2, bilinearly -> [B, sum C_i, 1, N]  (reference modules.py:24-54)."""

    def __init__(self, map_size=137):
        super().__init__()
        self.map_size = map_size
        self._caches = {}

    def prepared(self, img_featuremaps):
        dev = img_featuremaps[0].device.index          # DataParallel replicas share this dict: one slot per device
        return self._caches.setdefault(("img", dev), hotpath._Cache()).get(
            list(img_featuremaps),
            lambda: hip.prep_img_maps([m.detach().float() for m in img_featuremaps], self.map_size))

    def forward(self, img_featuremaps, pc, trans_mat):
        hotpath.require_hip(pc, "PerceptualPooling.forward(pc)")
        return hotpath.percep_pool(list(img_featuremaps), pc, trans_mat, self.prepared(img_featuremaps))

    def __repr__(self):
        return f"{self.__class__.__name__} (Map pc to {self.map_size} x {self.map_size} plane)"


class VoxelDecoder(nn.Module):
    """Parameters of the implicit MLP (fc.fc_0 .. fc.fc_out, Conv1d k=1) and the 7-point stencil
    (reference modules.py:193-214).  `displacments` keeps the reference's spelling and is a plain
    attribute, not a buffer, so it is absent from checkpoints exactly as in the reference."""

    def __init__(self, feature_size, h_dim):
        super().__init__()
        self.fc = nn.ModuleDict({
            "fc_0": nn.Conv1d(feature_size, h_dim * 2, 1),
            "fc_1": nn.Conv1d(h_dim * 2, h_dim, 1),
            "fc_2": nn.Conv1d(h_dim, h_dim, 1),
            "fc_out": nn.Conv1d(h_dim, 1, 1),
        })
        self.actvn = nn.ReLU()
        self.displacments = hotpath.stencil_offsets("cpu")
        self.precision = "bf16x3"
        self._caches = {}

    def mlp_params(self):
        return {f"{n}.{k}": getattr(self.fc[n], k) for n in ("fc_0", "fc_1", "fc_2", "fc_out")
                for k in ("weight", "bias")}

    # prepared copies (packed MLP planes, channels-last maps) are cached per device in self._caches; anything that
    # can change parameters behind the version counters drops them
    def invalidate(self):
        self._caches.clear()

    def train(self, mode=True):
        self.invalidate()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate()
        return super().load_state_dict(*args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):        # reached when a PARENT module loads a checkpoint
        self.invalidate()
        return super()._load_from_state_dict(*args, **kwargs)


class VoxelDecoder2(VoxelDecoder):
    """forward(p, feat, percep_feat) -> sdf [B,N]  (reference modules.py:255-282)."""

    def forward(self, p, feat, percep_feat):
        return hotpath.sdf_query(p, None, None, feat, self.mlp_params(), perm=(0, 1, 2), scale=1.0,
                                 precision=self.precision, percep_feat=percep_feat, caches=self._caches)

    def query(self, query, img_featuremaps, trans_mat, feat, map_size=137, perm=(2, 1, 0), scale=2.0,
              ordered_points=False, project_percep=None):
        """Fused PerceptualPooling + VoxelDecoder2 on RAW queries (reference models.py:91-97) without
        materialising the [B,1024,N] perceptual tensor.  ordered_points: see hotpath.sdf_query."""
        return hotpath.sdf_query(query, trans_mat, img_featuremaps, feat, self.mlp_params(), perm=perm,
                                 scale=scale, map_size=map_size, precision=self.precision,
                                 caches=self._caches, ordered_points=ordered_points, project_percep=project_percep)


# ======================================================================================= per-image modules
def _conv_bn_relu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class PointMLP(nn.Module):
    """Shared per-point MLP 3 -> 64 -> 256 -> 512 over the coarse cloud (reference modules.py:62-104)."""

    def __init__(self):
        super().__init__()
        self.block1 = _conv_bn_relu(3, 64)
        self.block2 = _conv_bn_relu(64, 256)
        self.block3 = _conv_bn_relu(256, 512)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = x.unsqueeze(3).permute(0, 2, 3, 1)          # [B,N,3] -> [B,3,1,N]
        return self.block3(self.block2(self.block1(x)))


class TreeGraphDecoder(nn.Module):
    """Tree-GCN stack: [B,1,F0] root feature -> [B, prod(degrees), 3] (reference modules.py:107-132)."""

    def __init__(self, batch_size, features, degrees, support):
        super().__init__()
        self.batch_size = batch_size
        self.layer_num = len(features) - 1
        assert self.layer_num == len(degrees), \
            "Number of features should be one more than number of degrees."
        self.pointcloud = None
        self.gcn = nn.Sequential()
        nodes = 1
        for i in range(self.layer_num):
            self.gcn.add_module(f"TreeGCN_{i}", TreeGCN(batch_size, i, features, degrees, support=support,
                                                        node=nodes, upsample=True,
                                                        activation=(i != self.layer_num - 1)))
            nodes *= degrees[i]

    def forward(self, tree):
        return self.gcn(tree)[-1]


class VoxelEncoder2(nn.Module):
    """3-D conv pyramid over the occupancy grid -> 6 feature volumes (reference modules.py:401-442):
    sigmoid head at full res, then conv-conv-BN blocks each followed by 2x max-pooling."""

    def __init__(self, layers):
        super().__init__()
        self.layers = layers
        self.conv = nn.ModuleDict()
        self.bn = nn.ModuleList()
        self.relu, self.sigmoid, self.maxpool = nn.ReLU(), nn.Sigmoid(), nn.MaxPool3d(2)
        for l in range(len(layers) - 1):
            self.conv[f"conv_{l}"] = nn.Conv3d(layers[l], layers[l + 1], 3, padding=1)
            if l > 2:
                self.conv[f"conv_{l}_0"] = nn.Conv3d(layers[l + 1], layers[l + 1], 3, padding=1)
            self.bn.append(nn.BatchNorm3d(layers[l + 1]))     # bn[2] exists but is unused, as in the reference

    def forward(self, x):
        net, out = x.unsqueeze(1), []
        for l in range(len(self.layers) - 1):
            if l < 2:
                net = self.bn[l](self.relu(self.conv[f"conv_{l}"](net)))
            elif l == 2:
                net = self.sigmoid(self.conv[f"conv_{l}"](net))
                out.append(net)
            else:
                net = self.relu(self.conv[f"conv_{l}"](net))
                net = self.bn[l](self.relu(self.conv[f"conv_{l}_0"](net)))
                out.append(net)
                net = self.maxpool(net)
        return out


class ResEncoder(nn.Module):
    """ResNet-18 with a stride-1 7x7 stem: global 128-vector + 5 maps at H, H/2, H/4, H/8, H/16
    (reference modules.py:1027-1074)."""

    def __init__(self):
        super().__init__()
        trunk = resnet18(pretrained=True)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=(7, 7), stride=(1, 1), padding=(3, 3), bias=False)
        self.bn1, self.relu, self.maxpool = trunk.bn1, trunk.relu, trunk.maxpool
        self.layer1, self.layer2, self.layer3, self.layer4 = (trunk.layer1, trunk.layer2, trunk.layer3,
                                                              trunk.layer4)
        self.avgpool, self.fc = trunk.avgpool, trunk.fc
        self.fc1 = nn.Linear(1000, 128)

    def forward(self, input_view):
        f0 = self.relu(self.bn1(self.conv1(input_view)))
        f1 = self.layer1(self.maxpool(f0))
        f2 = self.layer2(f1)
        f3 = self.layer3(f2)
        f4 = self.layer4(f3)
        vec = self.fc1(self.fc(torch.flatten(self.avgpool(f4), 1)))
        return vec, [f0, f1, f2, f3, f4]
