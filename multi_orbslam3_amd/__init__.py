"""MI355X-native ORB-SLAM3 hot path (ORB extractor, Hamming matchers, local BA) -- Python host mirror.

The compute lives in liborbgpu.so (hand-written HIP for gfx950 behind the C-ABI of include/orbgpu.h).
"""
__version__ = "0.1.0"
