"""Exposure start times of a visit, orbit by orbit.

Mirror of wayne/visit_planner.py:5-129 with plain floats (minutes): each HST
orbit (period 95 min) offers `time_per_orbit` of visibility after a guide-star
acquisition (6 min on the first orbit, 5 after); exposures of
`exptime + exp_overhead` follow one another, with a 5.8 min buffer dump after
every `num_exp_per_buffer + 1` exposures.
"""
import numpy as np


def VisitPlanner(detector, NSAMP, SAMPSEQ, SUBARRAY, num_orbits=3, time_per_orbit=54.0, hst_period=95.0,
                 exp_overhead=1.0):
    """:returns: dict with 'exp_times' (minutes from the visit start), 'orbit_start_index',
    'buffer_dump_index', 'num_exp', 'exptime' (s) and the inputs."""
    exptime = detector.exptime(NSAMP, SUBARRAY, SAMPSEQ)           # seconds
    exp_per_dump = detector.num_exp_per_buffer(NSAMP, SUBARRAY)
    time_buffer_dump = 5.8                                          # minutes (visit_planner.py:76)
    exp_times, orbit_start_index, buffer_dump_index = [], [], []
    for orbit_n in range(num_orbits):
        guide_star_aq = 6.0 if orbit_n == 0 else 5.0
        orbit_start_index.append(len(exp_times))
        start_time = hst_period * orbit_n
        visit_time = start_time + guide_star_aq
        visit_end_time = start_time + time_per_orbit
        exp_n = 0
        while visit_time < visit_end_time:
            exp_times.append(visit_time)
            visit_time += exptime / 60.0 + exp_overhead
            exp_n += 1
            if exp_n > exp_per_dump:
                visit_time += time_buffer_dump
                exp_n = 0
                buffer_dump_index.append(len(exp_times))
    return {"exp_times": np.array(exp_times), "NSAMP": NSAMP, "SAMPSEQ": SAMPSEQ, "SUBARRAY": SUBARRAY,
            "num_exp": len(exp_times), "exptime": exptime, "num_orbits": num_orbits, "exp_overhead": exp_overhead,
            "time_per_orbit": time_per_orbit, "hst_period": hst_period, "buffer_dump_index": buffer_dump_index,
            "orbit_start_index": orbit_start_index}
