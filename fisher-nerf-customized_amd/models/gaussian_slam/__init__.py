"""`models.gaussian_slam` import path named by the north star (commented out in the reference's tester, l.20/39)."""
from models.SLAM.gaussian import GaussianSLAM, FisherOps  # noqa: F401
from models.SLAM.gaussian_object import GaussianObjectSLAM  # noqa: F401


class PruneException(Exception):
    """models/utils.py:20 of the reference."""
    pass
