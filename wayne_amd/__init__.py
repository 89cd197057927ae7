"""wayne_amd -- MI355X (gfx950) WFC3-IR exposure synthesis.

A from-scratch replacement for the data-parallel hot path of
ucl-exoplanets/wayne (grism trace + double-gaussian electron thrower,
wavelength-dependent flat, sky / cosmic rays / gain per read, dark /
non-linearity / clipping / read noise per ramp) behind the reference's
``pyparallel.apply_psf`` and ``ExposureGenerator`` interfaces.  All compute
runs in hand-written HIP kernels reached through a C ABI
(include/wayne_hip.h); importing this package never falls back to the CPU.
"""
__version__ = "0.1.0"
