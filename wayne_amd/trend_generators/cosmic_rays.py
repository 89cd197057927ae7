"""Cosmic-ray generator parameters.

The reference builds a (size, size) frame of hits on the host per read
(wayne/trend_generators/cosmic_rays.py:88-139); here the hits are drawn and
scattered by the k_cosmic HIP kernel, so these classes only carry the
parameters that ExposureGenerator hands to the device.
"""


class BaseCosmicGenerator(object):
    full_frame_rate = 11       # hits per second per 1024 x 1024 (cosmic_rays.py:29)
    energy = 25000             # electrons (cosmic_rays.py:53)

    def _rate_full_frame_to_size(self, full_frame_rate, size):
        num_pixels = size * size if isinstance(size, int) else size[0] * size[1]
        return full_frame_rate / (1024. * 1024.) * num_pixels      # cosmic_rays.py:33-44


class MinMaxPossionCosmicGenerator(BaseCosmicGenerator):
    """Poisson number of hits at `rate` per second per full frame, energies
    uniform in [min_count, max_count) (cosmic_rays.py:106-139)."""

    def __init__(self, rate=11., min_count=10000, max_count=35000):
        if (min_count, max_count) != (10000, 35000):
            raise ValueError("the device kernel draws energies in [10000, 35000) as scanning_frame does "
                             "(exposure_generator.py:498-501)")
        self.rate = rate
        self.min_count = min_count
        self.max_count = max_count
