"""Generate tests/golden/roi.npz by RUNNING THE REFERENCE's preprocessing classes (src/datas/transform.py) on synthetic uint8 ROIs:

    python -m oracle.gen_golden_roi

Stores data only: per case the crc32 / float64 sum of the float32 result, a strided sample and the crop triple the reference drew.
Refuses to write if oracle/roi_ref.py is not bit-identical to the reference."""
import importlib.util
import os
import random
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"val96": ("val", 5, 96, 96, 11), "val_odd": ("val", 3, 101, 93, 12), "val88": ("val", 2, 88, 88, 13),
         "train96": ("train", 4, 96, 96, 14), "train_b": ("train", 4, 96, 96, 15), "train_odd": ("train", 3, 120, 99, 16)}


def main():
    sys.path.insert(0, ROOT)
    from oracle.ref_import import prepare_path

    prepare_path()
    spec = importlib.util.spec_from_file_location("ref_transform", "/root/reference/src/datas/transform.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)  # the reference module, loaded by path (its package __init__ needs soundfile / lightning)
    from oracle.roi_ref import preprocess, roi_inputs

    pipes = ref.get_preprocessing_pipelines()
    out = {}
    for name, (mode, T, H, W, seed) in CASES.items():
        frames = roi_inputs(T, H, W, seed)
        random.seed(seed)
        y_ref = pipes[mode](frames.copy()).astype(np.float32)
        y_or, crop = preprocess(frames, mode, rng=random.Random(seed))
        assert y_ref.shape == y_or.shape and np.array_equal(y_ref, y_or), name
        print(f"{name}: {mode} {T}x{H}x{W} -> {y_ref.shape} crop {crop}  bit-identical")
        out[f"{name}_cfg"] = np.array([T, H, W, seed, 1 if mode == "train" else 0])
        out[f"{name}_crop"] = np.array(crop)
        out[f"{name}_crc"] = np.array([zlib.crc32(np.ascontiguousarray(y_ref).tobytes())], dtype=np.uint64)
        out[f"{name}_sum"] = np.array([y_ref.astype(np.float64).sum()])
        out[f"{name}_sample"] = y_ref[:, ::9, ::7].copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "roi.npz"), **out)
    print("wrote tests/golden/roi.npz")


if __name__ == "__main__":
    main()
