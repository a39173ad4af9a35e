"""`gridencoder.grid_clustering` -- drop-in for the reference's gridencoder/grid_clustering.py.

`GridEncoder_clustering` (grid_clustering.py:93-217) is the same hash-grid encoder as `GridEncoder` plus a
per-level `ClusteringLayer` (:93-127) whose Student-t soft-assignment KL loss is evaluated on slices
`embeddings[offsets[i]:offsets[i+1]]` of the table.  The encode runs on the HIP kernels; the loss stays
plain torch, as in the reference (it only needs the table to remain sliceable by `offsets`).
"""
import numpy as np
import torch
import torch.nn as nn

from .grid import GridEncoder, _grid_encode, grid_encode  # noqa: F401  (the reference module carries its own copy of both)


class ClusteringLayer(nn.Module):
    def __init__(self, n_clusters=4, hidden=2, cluster_centers=None, alpha=1.0):
        super().__init__()
        self.n_clusters = n_clusters
        self.alpha = alpha
        self.hidden = hidden
        if cluster_centers is None:
            dev = "cuda" if torch.cuda.is_available() else "cpu"  # the reference hard-codes .cuda()
            cluster_centers = torch.zeros(n_clusters, hidden, dtype=torch.float, device=dev)
            cluster_centers.uniform_(-1e-4, 1e-4)
        self.cluster_centers = nn.Parameter(cluster_centers)
        self.kl_loss = nn.KLDivLoss(reduction="mean")

    def forward(self, x):
        # x [N, hidden] -> soft assignment [N, n_clusters] with a Student-t kernel
        dist2 = ((x.unsqueeze(1) - self.cluster_centers) ** 2).sum(2)
        q = (1.0 / (1.0 + dist2 / self.alpha)) ** (float(self.alpha + 1) / 2)
        return q / q.sum(dim=1, keepdim=True)

    @staticmethod
    def _target(q):
        p = (q ** 2) / q.sum(0)
        return (p / p.sum(dim=1, keepdim=True)).detach()

    def clustering_loss(self, x):
        q = self(x)
        return self.kl_loss(q.log(), self._target(q))


class GridEncoder_clustering(GridEncoder):
    def __init__(self, input_dim=3, num_levels=4, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__(input_dim, num_levels, level_dim, per_level_scale, base_resolution, log2_hashmap_size, desired_resolution,
                         gridtype, align_corners)
        self.cluster_layers = nn.ModuleList([ClusteringLayer() for _ in range(num_levels)])
        self.kl_loss = nn.KLDivLoss(reduction="mean")

    def clustering_loss(self, pick_level=True):
        levels = np.random.choice(np.arange(self.num_levels), [1]) if pick_level else np.arange(self.num_levels)
        offsets = self.offsets.tolist()
        loss = 0.0
        for i in levels:
            rows = self.embeddings[offsets[i]: offsets[i + 1]]
            q = self.cluster_layers[i](rows)
            loss = loss + self.kl_loss(q.log(), ClusteringLayer._target(q))
        return loss
