"""lvt_amd: MI355X-native implementation of the Latent Video Transformer hot path.

Host side mirrors the reference's (`vidgen`) registry / config / build_model surface; all compute
goes through hand-written gfx950 kernels in liblvt_hip.so (see include/lvt_hip.h).
"""
__version__ = "0.1"
