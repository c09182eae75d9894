"""Import stub (oracle/gen_golden_roi.py only): the reference's transform.py imports cv2 at module level and uses one call of it,
`cv2.flip(frame, 1)` = mirror the columns - a permutation, no arithmetic."""
import numpy as _np


def flip(img, code):
    assert code == 1
    return _np.ascontiguousarray(img[:, ::-1])
