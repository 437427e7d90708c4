"""distChamfer with the reference's name/signature (nnutils/chamfer_python.py:43-64), HIP-backed."""
from . import functional as UF


def distChamfer(a, b):
    """a [B,n,D], b [B,m,D] (D in {2,3}) -> (dist1 [B,n], dist2 [B,m], idx1 int32, idx2 int32):
    squared distance to the nearest point of the other set, expanded as |x|^2+|y|^2-2x.y like the
    reference, without materialising the [B,n,m] matrix."""
    return UF.chamfer(a, b)
