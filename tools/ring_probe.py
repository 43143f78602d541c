"""Developer probe: what the per-step audio hand-off to a side stream (engine.AudioGatherRing, the multi-GPU benchmark's
gather choreography) costs the pipelined step on one GPU."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from conan_amd.engine import AudioGatherRing
ctx, chp, vhp = bench.build_context(0)
eng, chunks = bench.make_engine(ctx, 64, 0)
st, slots, seg, hop = eng.st, eng.slots, eng.seg, ctx.hop
codes = torch.empty(64, seg, dtype=torch.int32, device='cuda'); mel = torch.empty(64, seg, 80, device='cuda')
def run(mode, nb=4, steps=60):
    ring = AudioGatherRing(lambda: torch.empty(64, seg * hop, device='cuda'), 1, 0, nb=nb, always=True)
    side = torch.cuda.Stream()
    def step(j):
        if mode == 'plain':
            st.step_async(slots, chunks[j % len(chunks)], ring.bufs[j % nb], emit=seg, codes=codes, mel_out=mel)
        elif mode == 'join_side':
            st.step_async(slots, chunks[j % len(chunks)], ring.bufs[j % nb], emit=seg, codes=codes, mel_out=mel)
            with torch.cuda.stream(side):
                st.join()
        elif mode == 'fence':
            buf, fence = ring.acquire(j, fence=True)
            st.step_async(slots, chunks[j % len(chunks)], buf, emit=seg, codes=codes, mel_out=mel, out_fence=fence)
            ring.submit(j, join=st.join)
        else:
            buf = ring.acquire(j)
            st.step_async(slots, chunks[j % len(chunks)], buf, emit=seg, codes=codes, mel_out=mel)
            ring.submit(j, join=st.join)
    for j in range(10): step(j)
    st.join(); ring.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(10, 10 + steps): step(j)
    st.join(); ring.drain(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for mode, nb in (('plain', 4), ('join_side', 4), ('ring', 4), ('fence', 4), ('fence', 2), ('plain', 4)):
    print("%-10s nb=%d  %.3f ms per step" % (mode, nb, run(mode, nb)))
main = torch.cuda.Stream()
with torch.cuda.stream(main):
    for mode, nb in (('plain', 4), ('ring', 4), ('ring', 8)):
        print("non-default stream: %-10s nb=%d  %.3f ms per step" % (mode, nb, run(mode, nb)))
