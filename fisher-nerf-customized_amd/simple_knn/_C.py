"""`simple_knn._C.distCUDA2(points[P,3]) -> [P]`: mean squared distance to the 3 nearest other points."""
from fisher_rast import ops as _ops


def distCUDA2(points):
    return _ops.knn_dist2(points)
