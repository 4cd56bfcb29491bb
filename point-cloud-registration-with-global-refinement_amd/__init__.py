"""MI355X-native pairwise point-cloud registration (FGR + multiscale GICP) behind the reference's
``ALL_FUNCTIONS.py`` call surface.  Compute lives in ``libpcr_hip.so`` (``csrc/``, C ABI in
``include/pcr_hip.h``); this package is the host-side mirror of the reference interface.

The directory name contains hyphens; import it with ``importlib.import_module`` or through the
``pcr_amd`` alias module at the repository root.
"""
from . import _lib, drivers, functions, geometry, io, o3d, posegraph, refinement, registration, sharding  # noqa: F401
from .functions import (Coarse_to_fine_FGR_M_GICP, GICP_robusto, Multiscale_GICP, amostragem_multiescala_otimizada, calculate_RMSE_and_fitness,  # noqa: F401
                        create_scales, radius_from_cloud_pair, registro_FGR, script1, script2)
from .geometry import (KDTreeSearchParamHybrid, KDTreeSearchParamKNN, KDTreeSearchParamRadius, PointCloud)  # noqa: F401

__version__ = "0.1.0"
