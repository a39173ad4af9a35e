from .grid import GridEncoder, grid_encode  # noqa: F401
from .grid_clustering import GridEncoder_clustering, ClusteringLayer  # noqa: F401
