import sys
sys.path.insert(0, "tests")
import test_hip_backward as t
for dtype in ("bf16x3", "f32"):
    for seed in (None, 1, 2, 3, 4):
        try:
            t._check_parameter_gradients(True, 2, 4096, 2, 6, dtype, seed)
            print(dtype, seed, "ok")
        except AssertionError as e:
            print(dtype, seed, "FAIL", str(e)[:100])
