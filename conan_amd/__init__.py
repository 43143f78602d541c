"""conan_amd: MI355X-native chunkwise streaming voice-conversion inference path
(Emformer -> Conan -> causal pixel-shuffle HiFi-GAN) behind the reference's
module/forward API.  Compute lives in csrc/ (HIP, gfx950) behind the C-ABI of
include/conan_hip.h; this package is the host-side mirror of the reference's
Python interfaces."""
__version__ = "0.1.0"

import os as _os

# HIP serves a process's streams from GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one are served in
# submission order.  Pipelined steps (conan_step_async) use three internal streams beside the caller's: any further
# stream of the application (a collective's side stream, a copy stream) then aliases one of them and a pipeline stage
# waits behind an unrelated event wait (measured: 1.85 -> 2.59 ms per step at 64 streams).  Only effective when this
# package is imported before the HIP runtime initialises; otherwise export the variable yourself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
