"""pixelspointspolygons_amd — MI355X-native (gfx950) encoder / fusion / decoder path of the P3 baselines.

Compute = hand-written HIP kernels in csrc/ behind the C-ABI of include/p3hip.h (libp3hip.so, loaded with
ctypes).  PyTorch-ROCm is used for device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
