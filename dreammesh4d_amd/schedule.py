"""Scheduled scalars of the configs: `C(value, epoch, global_step, interpolation)`
(threestudio/utils/misc.py:66-101), e.g. `lambda_mask: [200, 500., 5000., 1000]`
(custom/threestudio-dreammesh4d/configs/sugar_dynamic_dg.yaml:141).  Pinned by
tests/golden/schedule_C.npz (values computed by the reference function)."""
import math


def C(value, epoch, global_step, interpolation="linear"):
    if isinstance(value, (int, float)):
        return value
    v = list(value)
    if len(v) == 3:
        v = [0] + v
    if len(v) >= 6:                      # piecewise: [s0, v0, v1, s1, v2, s2, ...]
        sel = 3
        for i in range(3, len(v) - 2, 2):
            if global_step >= v[i]:
                sel = i + 2
        if sel != 3:
            start_value, start_step = v[sel - 3], v[sel - 2]
        else:
            start_step, start_value = v[:2]
        v = [start_step, start_value, v[sel - 1], v[sel]]
    if len(v) != 4:
        raise TypeError("scalar specification must be a number or a list of 3, 4 or >= 6 entries")
    start_step, start_value, end_value, end_step = v
    cur = global_step if isinstance(end_step, int) else epoch
    t = max(min(1.0, (cur - start_step) / (end_step - start_step)), 0.0)
    if interpolation == "linear":
        return start_value + (end_value - start_value) * t
    if interpolation == "exp":
        return math.exp(math.log(start_value) * (1 - t) + math.log(end_value) * t)
    raise ValueError(f"Unknown interpolation method: {interpolation}, only support linear and exp")
