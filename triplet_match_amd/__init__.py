"""triplet_match_amd — MI355X-native triplet-feature RANSAC registration hot path.

The product is the C-ABI library built from triplet_match_amd/csrc (HIP, gfx950) and declared in
include/tm_hip.h; this package holds its ctypes binding (capi), the build recipe (build) and the
seeded synthetic inputs (synth). There is no CPU fallback: importing capi without the built
library raises.
"""
__all__ = ["synth"]
