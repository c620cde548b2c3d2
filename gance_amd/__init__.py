"""
gance_amd: MI355X-native implementation of GANce's one hot path,
audio -> latent -> StyleGAN2 frame synthesis.

The package mirrors the reference's module names for that path only
(`network_interface`, `vector_sources`, `apply_spectrogram`,
`data_into_network_visualization`, `projection`, `overlay`) and sits on a C-ABI HIP library
(`gance_amd/csrc`, header `include/gance_hip.h`).
"""

__version__ = "0.1.0"
