"""Cosmic-ray generators on the host: the public classes of wayne/trend_generators/cosmic_rays.py.

The exposure path does not use them -- there the hits of a read interval are drawn on the device from Philox
counters by the same rule (`cosmic_hits`, wayne_amd/csrc/k_prep.h: Poisson number at `rate` per second per
1024 x 1024 pixels scaled to the frame, energy randint(min_count, max_count), pixel randint(0, size)) -- but user
code that builds cosmic frames itself imports these names, so they are kept, with the reference's method names and
arguments.  Random numbers: a `numpy.random.RandomState` given at construction (`rng=`), or numpy's global legacy
stream as in the reference.
"""
import numpy as np


class BaseCosmicGenerator(object):
    """11 cosmics per second of energy 25000, whatever the array size (cosmic_rays.py:10-107)."""

    def __init__(self, rng=None):
        self.rng = rng if rng is not None else np.random

    def _number_of_cosmics(self, time, size=1024):
        return int(11 * time)                                    # (:31-33; an int, so that a list can be built from it)

    def _rate_full_frame_to_size(self, full_frame_rate, size):
        """counts per full frame (1024, 1024) -> per `size` (int, square; or (height, width)) (:35-46)"""
        num_pixels = size * size if isinstance(size, (int, np.integer)) else size[0] * size[1]
        return full_frame_rate / (1024. * 1024.) * num_pixels

    def _generate_cosmic_energies(self, number):
        return [25000] * number                                  # (:48-55)

    def _generate_array(self, size):
        try:
            size = int(size)
            return np.zeros((size, size))
        except TypeError:
            return np.zeros(size)

    def _cosmics_to_array(self, list_of_energies, array):
        n = len(list_of_energies)
        y_pos = self.rng.randint(0, len(array), n)               # (:80-81)
        x_pos = self.rng.randint(0, len(array[0]), n)
        for i, e in enumerate(list_of_energies):
            array[y_pos[i], x_pos[i]] += e
        return array

    def cosmic_frame(self, time, size=1024):
        """A frame of cosmic-ray hits for `time` seconds on an array of `size` (:88-105)."""
        n = self._number_of_cosmics(time, size)
        return self._cosmics_to_array(self._generate_cosmic_energies(n), self._generate_array(size))


class MinMaxPossionCosmicGenerator(BaseCosmicGenerator):
    """Poisson number of hits at `rate` per second per 1024 x 1024 pixels, energies uniform in
    [min_count, max_count) (cosmic_rays.py:108-139)."""

    def __init__(self, rate=11., min_count=10000, max_count=35000, rng=None):
        BaseCosmicGenerator.__init__(self, rng)
        self.rate, self.min_count, self.max_count = rate, min_count, max_count

    def _number_of_cosmics(self, time, size=1024):
        return self.rng.poisson(self._rate_full_frame_to_size(self.rate, size) * time)

    def _generate_cosmic_energies(self, number):
        return self.rng.randint(self.min_count, self.max_count, number)
