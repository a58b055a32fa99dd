"""cobevt_amd — MI355X-native Fused Axial Attention (FAX) hot path of CoBEVT (SinBEVT / FuseBEVT / CoBEVT).

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) calling hand-written HIP kernels
for gfx950 through a C ABI (include/cobevt_hip.h, cobevt_amd/csrc).  `cobevt_amd.host` mirrors the reference's
nn.Module interface; `cobevt_amd.registry.create_model` mirrors its model registry.
"""
__version__ = "0.1.0"
