"""conan_amd: MI355X-native chunkwise streaming voice-conversion inference path
(Emformer -> Conan -> causal pixel-shuffle HiFi-GAN) behind the reference's
module/forward API.  Compute lives in csrc/ (HIP, gfx950) behind the C-ABI of
include/conan_hip.h; this package is the host-side mirror of the reference's
Python interfaces."""
__version__ = "0.1.0"
