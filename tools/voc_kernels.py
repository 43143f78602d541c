"""Per-kernel times of the vocoder step at B streams for a list of developer plans (conan_streams_opts.dev_plan), one process:
    python tools/voc_kernels.py 64 "" "UPS_CFG=1" "UPS_CFG=0;..."
HIP events around every matrix-kernel launch (conan_profile_begin / _end), blocking hifigan steps, and the step time of the launches
back to back without profiling."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

from conan_amd import configs, synth  # noqa: E402
from conan_amd.runtime import Context  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    plans = sys.argv[2:] or [""]
    frames = int(os.environ.get("VK_FRAMES", "4"))
    arith = os.environ.get("VK_ARITH", "auto")
    vhp = configs.hifigan_hparams()
    ctx = Context(None, vhp, 0, False, False, True)
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    mel = torch.rand(B, frames, 80, device="cuda") * 4 - 5
    ids = list(range(B))
    for plan in plans:
        st = ctx.streams(B, max_frames=frames, max_ref_frames=16, arith=arith, dev_plan=plan or None)
        st.reset(ids)
        wav = torch.empty(B, frames * ctx.hop, device="cuda")
        for _ in range(10):
            st.hifigan_step(ids, mel, out=wav)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 60
        for _ in range(n):
            st.hifigan_step(ids, mel, out=wav)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        st.profile_begin()
        for _ in range(5):
            st.hifigan_step(ids, mel, out=wav)
        st.profile_end()
        print(f"== plan '{plan}'  B={B} frames={frames} arith={st.arith}: {ms:.4f} ms per vocoder step (back to back)")
        for name, kms, fl, cnt in sorted(st.profile_kernels(), key=lambda r: -r[1]):
            print(f"   {kms * 1e3 / cnt:8.1f} us x {cnt / 5:4.1f}/step  {fl / (kms * 1e-3) / 1e12:7.1f} TFLOP/s  {name}")
        st.close()
    ctx.close()


if __name__ == "__main__":
    main()
