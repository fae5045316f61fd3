"""Tree-structured graph convolution used by the coarse point decoder (per-image work, PyTorch).

Parameter names follow the reference (layers/gcn.py:6-69) so its checkpoints load:
W_root.{i}.weight, W_branch, W_loop.{0,1}.weight, bias."""
import math

import torch
import torch.nn as nn


class TreeGCN(nn.Module):
    def __init__(self, batch, depth, features, degrees, support=10, node=1, upsample=False,
                 activation=True):
        super().__init__()
        self.batch, self.depth = batch, depth
        self.in_feature, self.out_feature = features[depth], features[depth + 1]
        self.node, self.degree = node, degrees[depth]
        self.upsample, self.activation = upsample, activation

        self.W_root = nn.ModuleList(nn.Linear(features[i], self.out_feature, bias=False)
                                    for i in range(depth + 1))
        if upsample:
            self.W_branch = nn.Parameter(torch.empty(node, self.in_feature, self.degree * self.in_feature))
        self.W_loop = nn.Sequential(nn.Linear(self.in_feature, self.in_feature * support, bias=False),
                                    nn.Linear(self.in_feature * support, self.out_feature, bias=False))
        self.bias = nn.Parameter(torch.empty(1, self.degree, self.out_feature))
        self.leaky_relu = nn.LeakyReLU(negative_slope=0.2)
        self.reset_parameters()

    def reset_parameters(self):
        if self.upsample:
            nn.init.kaiming_normal_(self.W_branch.data, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        bound = 1.0 / math.sqrt(self.out_feature)
        self.bias.data.uniform_(-bound, bound)

    def forward(self, tree):
        leaves = tree[-1]
        batch = leaves.size(0)
        # ancestors: every level's projection, repeated down to this level's node count
        root = 0
        for level, w in zip(tree[: self.depth + 1], self.W_root):
            reps = self.node // level.size(1)
            root = root + w(level).repeat(1, 1, reps).view(batch, -1, self.out_feature)
        if self.upsample:
            grown = self.leaky_relu(leaves.unsqueeze(2) @ self.W_branch)
            grown = self.W_loop(grown.view(batch, self.node * self.degree, self.in_feature))
            out = root.repeat(1, 1, self.degree).view(batch, -1, self.out_feature) + grown
        else:
            out = root + self.W_loop(leaves)
        if self.activation:
            out = self.leaky_relu(out + self.bias.repeat(1, self.node, 1))
        tree.append(out)
        return tree
