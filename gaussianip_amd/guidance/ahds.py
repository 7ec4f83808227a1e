"""AHDS (annealed) timestep schedule.

Reference: threestudio/models/guidance/ipa_guidance.py — constants :200-210, dual-Gaussian fit :544-587
(L-BFGS-B on the squared error of three range masses), inverse-CDF table :590-599 (Nelder-Mead per step), per-step
sampling windows :625-638.  Pinned by tests/golden/ahds_schedule.npz (captured from the imported reference functions).

The fit and the table use the same scipy optimisers with the same arguments as the reference (scipy is the pinned
algorithm here, not re-implemented); only the objective's suffix sums are cached (same values, O(1) per call), which
cuts the ~8 s start-up cost of the reference to well under a second.
"""
import numpy as np
import torch
from scipy.optimize import minimize

AHDS_N = 2400
AHDS_T0 = 799
AHDS_MAX_T = 800
AHDS_TARGET_MASS = (0.41, 0.21, 0.375)
AHDS_RANGES = ((0, 350), (350, 450), (450, 800))
AHDS_INIT = (260, 60, 280)   # the reference's trailing comma (:206) makes this a 2-D x0; scipy >= 1.11 needs it 1-D
AHDS_BOUNDS = ((200, 400), (20, 100), (100, 300))


def _dual_gaussian(T, s1, s2, max_t):
    w = np.array([np.exp(-(t - T) ** 2 / (2 * s1 ** 2)) if t <= T else np.exp(-(t - T) ** 2 / (2 * s2 ** 2))
                  for t in range(max_t)])
    return w / np.sum(w)


def _mass_error(params, target, ranges, max_t):
    w = _dual_gaussian(params[0], params[1], params[2], max_t)
    return sum((np.sum(w[a:b]) - m) ** 2 for (a, b), m in zip(ranges, target))


def optimized_dual_gaussian(init=AHDS_INIT, target=AHDS_TARGET_MASS, ranges=AHDS_RANGES, max_t=AHDS_MAX_T,
                            bounds=AHDS_BOUNDS):
    """pdf over t in [0, max_t): left / right half-Gaussians around T fitted so that the three ranges carry `target` mass."""
    res = minimize(_mass_error, list(init), args=(target, ranges, max_t), bounds=list(bounds), method="L-BFGS-B")
    return _dual_gaussian(res.x[0], res.x[1], res.x[2], max_t)


def timestep_table(pdf, N=AHDS_N, t0=AHDS_T0):
    """chosen_t[i] = argmin_t | sum(pdf[t:]) - i/N | by Nelder-Mead started at t0, truncated to int (>= 0)."""
    tail = [sum(pdf[t:]) for t in range(len(pdf))]      # same left-to-right Python sums as the reference objective
    last = len(pdf) - 1

    def objective(t, i):
        return abs(tail[int(max(0, min(last, float(t[0]) if hasattr(t, "__len__") else t)))] - i / N)

    table = []
    for i in range(N):
        r = minimize(objective, t0, args=(i,), method="Nelder-Mead")
        table.append(max(0, int(r.x[0])))
    return table


class AHDSSchedule:
    """Holds the 2400-entry table and draws the per-step timesteps (ipa_guidance.py:625-638)."""

    def __init__(self, table=None):
        self.table = list(table) if table is not None else timestep_table(optimized_dual_gaussian())
        self.t_min = next((t for t in reversed(self.table) if t != 0), None)

    def window(self, step):
        """[lo, hi) of torch.randint for this step."""
        cur_t = self.table[step]
        if 0 <= step < 700:
            return 500, 800
        if 700 <= step < 900:
            return 400, cur_t + 50
        if 900 <= step < 1400:
            return 150, cur_t + 50
        return (20, cur_t + 50) if cur_t != 0 else (20, self.t_min)

    def sample(self, step, batch_size, device, generator=None):
        lo, hi = self.window(step)
        from .sds import per_sample
        return per_sample(lambda k, g: torch.randint(lo, hi, [k], dtype=torch.long, device=device, generator=g), batch_size, generator)
