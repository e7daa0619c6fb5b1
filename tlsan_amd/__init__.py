"""tlsan_amd: MI355X-native TLSAN hot path (HIP/gfx950 kernels behind the reference's Model surface)."""
__version__ = "0.1.0"
