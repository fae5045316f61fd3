"""CoarseNet and LIST with the reference's API (network/models.py:14-112): constructor takes the
config namespace, `LIST.forward(img, query, trans_mat=None) -> (vox_feat[0], sdf)`, same sub-module
attribute names (executors and train.py reach them through `.module`)."""
import torch
import torch.nn as nn

from . import modules as M


class CoarseNet(nn.Module):
    """RGB image -> coarse point cloud [B, prod(point_degree), 3]."""

    def __init__(self, config):
        super().__init__()
        self.image_encoder = M.ResEncoder()
        self.point_decoder = M.TreeGraphDecoder(config.train_batch_size, config.point_feat,
                                                config.point_degree, 10)

    def forward(self, rgba):
        featvecs, _ = self.image_encoder(rgba)
        return self.point_decoder([featvecs.unsqueeze(1)])


class LIST(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.vox_res = config.vox_res
        self.bb_min, self.bb_max = config.bb_min, config.bb_max
        enc_feat_size = sum(config.im_enc_layers[3:]) * 7 + 1024 + 3          # 3610

        # MI355X-first producer layout: the encoders whose maps feed the query path run
        # channels-last, so their outputs already have the [..spatial..][C] layout the gathers read and
        # list_prep_vox_maps is a no-op (MIOpen NDHWC/NHWC convolutions: same results, same speed).
        self.channels_last = bool(getattr(config, "channels_last", True))
        self.vox_encoder_half = getattr(config, "vox_encoder_precision", "fp32") == "fp16"
        self.vox_encoder = M.VoxelEncoder2(config.im_enc_layers)
        self.sdf_decoder = M.VoxelDecoder2(enc_feat_size, 256)
        self.sdf_decoder.precision = getattr(config, "precision", "bf16x3")
        self.percep_pooling = M.PerceptualPooling()
        self.im_encoder = M.ResEncoder()
        self.im_encoder2 = M.ResEncoder()
        self.point_decoder = M.TreeGraphDecoder(config.train_batch_size, config.point_feat,
                                                config.point_degree, 10)
        self.point_mlp_coarse = M.PointMLP()
        self.spatial_transformer = nn.Sequential(
            nn.Linear(128 + 512, 128), nn.LeakyReLU(0.2), nn.BatchNorm1d(128),
            nn.Linear(128, 128), nn.LeakyReLU(0.2), nn.BatchNorm1d(128),
            nn.Linear(128, 12))

    # ---- per-image stage (encoders, coarse cloud, camera, voxel pyramid) ---------------------------
    def _apply_memory_format(self, img):
        if self.channels_last and img.is_cuda:
            if not getattr(self, "_cl_done", False):
                self.vox_encoder.to(memory_format=torch.channels_last_3d)
                self.im_encoder2.to(memory_format=torch.channels_last)
                self._cl_done = True
            return img.contiguous(memory_format=torch.channels_last), True
        return img, False

    def encode(self, img, trans_mat=None):
        feat_g, _ = self.im_encoder(img)
        img_cl, use_cl = self._apply_memory_format(img)
        feat_g2, feat_l2 = self.im_encoder2(img_cl)
        pc = self.point_decoder([feat_g.unsqueeze(1)])
        coarse = torch.max(self.point_mlp_coarse(pc), -1)[0].reshape(img.shape[0], -1)
        if trans_mat is None:
            code = torch.cat([coarse, feat_g2.reshape(img.shape[0], -1)], dim=1)
            trans_mat = self.spatial_transformer(code).reshape(-1, 4, 3)
        occ = self.create_occ(pc)
        if self.vox_encoder_half and occ.is_cuda:
            with torch.autocast("cuda", dtype=torch.float16):
                vox_feat = self.vox_encoder(occ)
        else:
            vox_feat = self.vox_encoder(occ)
        if use_cl:      # MIOpen keeps the format; levels that lost it would simply be transposed again
            vox_feat = [v if v.shape[1] == 1 else v.contiguous(memory_format=torch.channels_last_3d)
                        for v in vox_feat]
        return feat_l2, vox_feat, trans_mat, pc, occ

    # ---- per-point stage: the HIP hot path ------------------------------------------------------------
    def query_sdf(self, query, feat_l2, vox_feat, trans_mat, ordered_points=False, project_percep=None):
        return self.sdf_decoder.query(query, feat_l2, trans_mat, vox_feat,
                                      map_size=self.percep_pooling.map_size, ordered_points=ordered_points,
                                      project_percep=project_percep)

    def forward(self, img, query, trans_mat=None):
        feat_l2, vox_feat, trans_mat, _, _ = self.encode(img, trans_mat)
        sdf = self.query_sdf(query, feat_l2, vox_feat, trans_mat)
        return vox_feat[0].float(), sdf                     # (a no-op unless the 3-D encoder ran in half precision)

    def create_occ(self, pc):
        """Voxelise the coarse cloud: nearest node of the regular bb grid gets 1.  The reference asks
        a host KD-tree built over the grid nodes (models.py:102-112, a device->host sync per step);
        for a regular grid the nearest node is a rounding, done here on the device.  Flat index is
        i*res^2 + j*res + k with i along x (meshgrid 'ij', utils.py:84-95)."""
        res = self.vox_res
        t = (pc.detach() - self.bb_min) / (self.bb_max - self.bb_min) * (res - 1)
        ijk = torch.clamp(torch.floor(t + 0.5), 0, res - 1).long()
        flat = (ijk[..., 0] * res + ijk[..., 1]) * res + ijk[..., 2]
        occ = torch.zeros((pc.shape[0], res ** 3), dtype=torch.float32, device=pc.device)
        occ.scatter_(1, flat, 1.0)
        return occ.view(pc.shape[0], res, res, res)
