"""umr_amd -- MI355X-native render-and-compare path for UMR-style mesh reconstruction.

Public surface mirrors the reference modules it replaces:
    umr_amd.smr.SoftRenderer            <- nnutils/smr.py
    umr_amd.functional.soft_rasterize   <- external/SoftRas/soft_renderer/functional/soft_rasterize.py
    umr_amd.loss_utils.*                <- nnutils/loss_utils.py, external/SoftRas/soft_renderer/losses.py
    umr_amd.chamfer_python.distChamfer  <- nnutils/chamfer_python.py
    umr_amd.geom_utils.*                <- nnutils/geom_utils.py
All of them call libumr_hip.so (include/umr_hip.h) and raise if it is missing -- there is no CPU path.
"""
__version__ = "0.1"
