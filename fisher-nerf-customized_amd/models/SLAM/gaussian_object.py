"""`GaussianObjectSLAM`: the object-aware variant's Fisher surface (models/SLAM/gaussian_object.py:1541-1551,
1591-1617, 1940-2045): 11 Fisher columns [mean xyz | opacity | scale xyz | rot rxyz], optional random Gaussians
appended with colour 0.5, `compute_Hessian(..., return_pose=True)` -> (cur_H, eye(6), vis_count)."""
from models.SLAM.gaussian import FisherOps, GaussianSLAM


class ObjectFisherOps(FisherOps):
    FISHER_COLUMNS = 11


class GaussianObjectSLAM(ObjectFisherOps, GaussianSLAM):
    pass
