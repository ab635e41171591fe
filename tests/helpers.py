"""Shared helpers of the parity tests."""
import numpy as np


def ulp_distance(a, b):
    """Element-wise distance in units in the last place between two float32 arrays
    (sign-magnitude aware; NaN never equal)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    d[np.isnan(a) | np.isnan(b)] = 1 << 40
    return d


def compare_frames(gpu_pp, gpu_ids, gpu_rgb, ora_pp, ora_ids, ora_rgb):
    """Returns a dict of parity figures between an engine frame and an oracle frame."""
    ulp = ulp_distance(gpu_pp[..., :3], ora_pp[..., :3])
    depth = ulp_distance(gpu_pp[..., 3], ora_pp[..., 3])
    return {
        "max_ulp": int(ulp.max()),
        "pixels_over_1ulp": int((ulp.max(axis=-1) > 1).sum()),
        "pixels_nonzero_ulp": int((ulp.max(axis=-1) > 0).sum()),
        "depth_max_ulp": int(depth.max()),
        "ids_equal": bool(np.array_equal(gpu_ids[..., :2], ora_ids[..., :2])),
        "ids_all_equal": bool(np.array_equal(gpu_ids, ora_ids)),
        "rgb_max_diff": int(np.abs(gpu_rgb.astype(int) - ora_rgb.astype(int)).max()),
        "rgb_equal": bool(np.array_equal(gpu_rgb, ora_rgb)),
    }
