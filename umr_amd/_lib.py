"""ctypes binding of libumr_hip.so (include/umr_hip.h).  No fallback: if the library is missing or a
call is rejected this raises -- the product path never routes through a CPU implementation."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libumr_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float
_Z = ctypes.c_size_t

# name -> argtypes, exactly the prototypes of include/umr_hip.h
SIGNATURES = {
    "umr_profile_enable": ([_I], _I),
    "umr_debug_set": ([ctypes.c_char_p, _I], _I),
    "umr_profile_collect": ([_I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long),
                             ctypes.POINTER(ctypes.c_double)], _I),
    "umr_raster_workspace_bytes": ([_I, _I], _Z),
    "umr_raster_workspace_bytes_for": ([_I, _I, _I], _Z),
    "umr_raster_state_bytes": ([_I, _I], _Z),
    "umr_raster_forward": ([_P] * 9 + [_I, _I, _I, _I, _F, _F, _F, _F, _I, _F, _F, _I, _I, _I, _I, _I,
                            ctypes.POINTER(ctypes.c_float), _P, _Z, _P], _I),
    "umr_raster_forward_vis": ([_P] * 9 + [_I, _I, _I, _I, _F, _F, _F, _F, _I, _F, _F, _I, _I, _I, _I, _I,
                                ctypes.POINTER(ctypes.c_float), _P, _Z, _P, _P], _I),
    "umr_raster_backward": ([_P] * 8 + [_I, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _I, _F, _F, _I, _I, _I, _I, _P, _Z, _P], _I),
    "umr_project_faces_forward": ([_P] * 5 + [_I, _I, _I, _F, _F, _I, _P], _I),
    "umr_project_faces_lit_forward": ([_P] * 6 + [_I, _I, _I, _F, _F, _I, _F, _F, ctypes.POINTER(_F), ctypes.POINTER(_F), _P], _I),
    "umr_project_faces_lit_backward": ([_P] * 9 + [_I, _I, _I, _I, _F, ctypes.POINTER(_F), ctypes.POINTER(_F), _P, _Z, _P], _I),
    "umr_project_workspace_bytes": ([_I, _I], _Z),
    "umr_project_faces_backward": ([_P] * 7 + [_I, _I, _I, _I, _P, _Z, _P], _I),
    "umr_rotate_cam_y": ([_P, _P, _P, _I, _P], _I),
    "umr_rotate_cam_axis": ([_P, _P, ctypes.POINTER(_F), _P, _I, _P], _I),
    "umr_project_points_forward": ([_P] * 3 + [_I, _I, _I, _F, _P], _I),
    "umr_project_points_backward": ([_P] * 5 + [_I, _I, _I, _P], _I),
    "umr_neg_iou_sums_stride": ([_L], _L),
    "umr_neg_iou_forward": ([_P, _L, _P, _P, _P, _Z, _I, _L, _P], _I),
    "umr_neg_iou_backward": ([_P, _L, _P, _P, _P, _P, _L, _I, _L, _P], _I),
    "umr_chamfer_forward": ([_P] * 6 + [_I, _I, _I, _I, _P], _I),
    "umr_chamfer_backward": ([_P] * 8 + [_I, _I, _I, _I, _P], _I),
    "umr_grid_sample_forward": ([_P] * 3 + [_I, _I, _I, _I, _L, _P], _I),
    "umr_grid_sample_backward": ([_P] * 5 + [_I, _I, _I, _I, _L, _P], _I),
    "umr_laplacian_forward": ([_P] * 5 + [_I, _I, _P], _I),
    "umr_laplacian_backward": ([_P] * 5 + [_I, _I, _P], _I),
    "umr_flatten_forward": ([_P] * 3 + [_I, _I, _I, _P], _I),
    "umr_flatten_backward": ([_P] * 4 + [_I, _I, _I, _P], _I),
    "umr_visible_face_mask": ([_P, _P, _I, _L, _I, _P], _I),
    "umr_upsample2x_bilinear_forward": ([_P, _P, _L, _I, _I, _P], _I),
    "umr_upsample2x_bilinear_backward": ([_P, _P, _L, _I, _I, _P], _I),
    "umr_cos_sim_workspace_bytes": ([_I, _I, ctypes.POINTER(_I)], _Z),
    "umr_cos_sim_forward": ([_I, ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _F, _P, _P,
                             _Z, _P], _I),
    "umr_cos_sim_backward": ([_I, ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_P),
                              ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _F, _P, _P, _Z, _P], _I),
    "umr_perceptual_prologue_forward": ([_P, _P, _P, _I, _I, ctypes.c_long, ctypes.POINTER(_F), ctypes.POINTER(_F), _P], _I),
    "umr_perceptual_prologue_backward": ([_P, _P, _P, _P, _P, _I, _I, ctypes.c_long, ctypes.POINTER(_F), _P], _I),
    "umr_part_match_workspace_bytes": ([_I, _I, _I], _Z),
    "umr_part_match_forward": ([_P, _P, _P, _I, _I, _I, ctypes.POINTER(_F), _F, _F, _P, _P, _P, _Z, _P], _I),
    "umr_part_match_backward": ([_P, _P, _P, _I, _I, _I, ctypes.POINTER(_F), _F, _F, _P, _P, _P, _P, _P, _Z, _P], _I),
    "umr_dt_barrier_workspace_bytes": ([_I, _I, _I], _Z),
    "umr_dt_barrier": ([_P, _P, _P, _P, _I, _I, _I, _F, _P, _Z, _P], _I),
    "umr_texture_atlas_shape": ([_I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)], _I),
    "umr_texture_atlas": ([_P, _P, _P, _P, _I, _I, _I, _F, _P], _I),
    "umr_kp_flow_workspace_bytes": ([_I, _I, _I], _Z),
    "umr_kp_flow_transfer": ([_P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _P, _Z, _P], _I),
    "umr_kp_cam_transfer": ([_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P], _I),
    "umr_reg_scratch_floats": ([_L, _I], _L),
    "umr_row_norm_mean_forward": ([_P, _P, _P, _Z, _L, _I, _P], _I),
    "umr_row_norm_mean_backward": ([_P, _P, _P, _L, _I, _P], _I),
    "umr_abs_mean_forward": ([_P, _P, _P, _Z, _L, _I, _I, _P], _I),
    "umr_abs_mean_backward": ([_P, _P, _P, _L, _I, _I, _P], _I),
    "umr_masked_l1_forward": ([_P, _P, _P, _P, _P, _P, _Z, _I, _I, _L, _P], _I),
    "umr_masked_l1_backward": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _P], _I),
}

_lib = None


def lib():
    """Load libumr_hip.so once.  Raises if it has not been built (python -m umr_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "umr_amd: %s is missing -- build it with `python -m umr_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        h.umr_version.restype = ctypes.c_char_p
        h.umr_build_id.restype = ctypes.c_char_p
        for name, (argtypes, restype) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = h
        # debugging / A-B aid: UMR_DEBUG_SET="key=value,key=value" applies umr_debug_set switches at load (tools/gpu_*.sh)
        for kv in os.environ.get("UMR_DEBUG_SET", "").split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                if h.umr_debug_set(k.strip().encode(), int(v)) != 0:
                    raise RuntimeError("umr_amd: UMR_DEBUG_SET names an unknown switch: %r" % kv)
    return _lib


def version():
    return lib().umr_version().decode()


def build_id():
    return lib().umr_build_id().decode()


def on_device(t):
    """True for a tensor this library can take: a CUDA (ROCm) tensor.  Every entry of the Python layer checks through here."""
    return t.is_cuda


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be CUDA (ROCm) and contiguous."""
    if t is None:
        return None
    if not on_device(t):
        raise RuntimeError("umr_amd: expected a GPU tensor, got %s (no CPU path exists)" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("umr_amd: tensor must be contiguous")
    return t.data_ptr()


def stream_ptr(device=None):
    """Raw handle of the current HIP stream of `device`.  Kernels are launched in the CURRENT device context (as the
    reference's were: no device guard), so a tensor on another GPU is rejected here
    instead of faulting inside the launch -- one process per GPU calls torch.cuda.set_device(local_rank) once.
    Uses torch's C accessors: torch.cuda.current_stream() costs ~8 us of Python per call, and the render-and-compare
    path makes ~30 C-ABI calls per step."""
    cur = torch._C._cuda_getDevice()
    idx = cur if device is None else torch.device(device).index
    if idx is None:
        idx = cur
    if idx != cur:
        raise RuntimeError("umr_amd: tensors live on cuda:%d but the current device is cuda:%d; call "
                           "torch.cuda.set_device first" % (idx, cur))
    return torch._C._cuda_getCurrentRawStream(idx)


# Measurement aid: when set to a callable, the raster operators hand it (name, dict of their input tensors) right before their
# C-ABI call -- how bench.py --capture-scene freezes the geometry the training step really renders (profiles/scenes/).  None in
# every other run; the operators do nothing else with it.
TAP = None

_TRACE = bool(os.environ.get("UMR_TRACE_SYNC"))   # debugging aid: name every C-ABI call and synchronise after it


def check(rc, what):
    if _TRACE:
        print("[umr] %s rc=%d" % (what, rc), flush=True)
        torch.cuda.synchronize()
    if rc != 0:
        raise RuntimeError("umr_amd: %s failed with status %d (%s)" % (
            what, rc, {-1: "rejected arguments / unsupported mode", -2: "kernel launch error"}.get(rc, "?")))


def profile_enable(on=True):
    check(lib().umr_profile_enable(1 if on else 0), "umr_profile_enable")


def profile_collect(which):
    """-> (total_ms, launches, algorithmic_bytes) of the raster forward (0) / backward (1) main kernel."""
    ms, n, b = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
    check(lib().umr_profile_collect(which, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(b)), "umr_profile_collect")
    return ms.value, n.value, b.value


def debug_set(key, value):
    check(lib().umr_debug_set(key.encode(), int(value)), "umr_debug_set(%s)" % key)
