"""Visit-long trends: one scale factor per exposure.

Mirror of wayne/trend_generators/visit_trends.py:10-73.  Pure host arithmetic
(one scalar per exposure); the factor enters the device path as
wayne_exposure_desc.scale_factor (exposure_generator.py:620-621).
"""
import numpy as np


def gen_orbit_start_times_per_exp(time_array, obs_start_index):
    """For every exposure, the start time of the orbit it belongs to (visit_trends.py:60-73)."""
    time_array = np.asarray(time_array, dtype=float)
    bounds = list(obs_start_index) + [len(time_array)]
    t_0 = np.zeros(len(time_array))
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        t_0[lo:hi] = time_array[lo]
    return t_0


class BaseVisitTrend(object):
    """Takes the visit planner's output ({'exp_start_times', 'orbit_start_index'})
    and produces one scaling factor per exposure; subclasses implement
    `_gen_scaling_factors`."""

    def __init__(self, visit_plan, coeffs=None):
        self.visit_plan = visit_plan
        self.coeffs = coeffs
        self.scale_factors = self._gen_scaling_factors(visit_plan, coeffs)

    def _gen_scaling_factors(self, visit_plan, coeffs):
        raise NotImplementedError

    def get_scale_factor(self, exp_num):
        return self.scale_factors[exp_num]


class HookAndLongTermRamp(BaseVisitTrend):
    def _gen_scaling_factors(self, visit_plan, coeffs):
        t = np.asarray(visit_plan["exp_start_times"], dtype=float)
        t_0 = gen_orbit_start_times_per_exp(t, visit_plan["orbit_start_index"])
        return self.ramp_model(t, t_0, *coeffs)

    @staticmethod
    def ramp_model(t, t_0, a1, b1, b2, to):
        """(1 - a1 (t - to)) (1 - b1 exp(-b2 (t - t_0))): long-term slope times
        the per-orbit exponential hook (visit_trends.py:44-57)."""
        t = np.asarray(t, dtype=float)
        return (1 - a1 * (t - to)) * (1 - b1 * np.exp(-b2 * (t - t_0)))
