"""detectandtrack_amd — MI355X-native hot path of facebookresearch/DetectAndTrack.

Host-side mirror of the reference's `lib/` package surface (core.config, modeling.*,
core.test, core.tracking_engine, utils.*) over hand-written HIP kernels reached through
the C ABI in include/dat_hip.h (detectandtrack_amd/libdat.py).
"""
__version__ = '0.1.0'
