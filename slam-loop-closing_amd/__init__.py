"""slam-loop-closing_amd — MI355X-native drop-in for the ORB/Hamming loop-closure hot path of
F-Fer/SLAM-Loop-Closing (LoopClosingSystem::matchFeatures / detectLoops, include/loop_closing.hpp:40,48).

The product is `lib/liblcm_hip.so` (C ABI in include/lcm.h, kernels in csrc/lcm_kernels.hip).  This Python package
is the ctypes plumbing used by tests, bench.py and __graft_entry__; the directory name has a hyphen, so it is
loaded by path (see `load_package` in __graft_entry__.py) under the module name `slam_loop_closing_amd`.
"""
from . import capi, synth, sharding  # noqa: F401
from .capi import Matcher, Group, LoopClosingSystem, LcmError, default_params, load_library  # noqa: F401

__all__ = ["capi", "synth", "sharding", "Matcher", "Group", "LoopClosingSystem", "LcmError", "default_params", "load_library"]
