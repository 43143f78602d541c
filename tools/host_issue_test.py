"""Developer probe: host time of each pipelined step call (is the host ahead of the device?)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
ctx, chp, vhp = bench.build_context(0)
eng, chunks = bench.make_engine(ctx, 64, 0)
seg, hop = eng.seg, ctx.hop
bufs = [torch.empty(64, seg * hop, device='cuda') for _ in range(4)]
codes = torch.empty(64, seg, dtype=torch.int32, device='cuda'); mel = torch.empty(64, seg, 80, device='cuda')
for j in range(6):
    eng.st.step_async(eng.slots, chunks[j % len(chunks)], bufs[j % 4], emit=seg, codes=codes, mel_out=mel)
eng.st.join(); torch.cuda.synchronize()
ts = [time.perf_counter()]
for j in range(6, 26):
    eng.st.step_async(eng.slots, chunks[j % len(chunks)], bufs[j % 4], emit=seg, codes=codes, mel_out=mel)
    ts.append(time.perf_counter())
eng.st.join(); torch.cuda.synchronize()
te = time.perf_counter()
print("host ms per call:", " ".join("%.2f" % ((b - a) * 1e3) for a, b in zip(ts[:-1], ts[1:])))
print("issue done at %.2f ms, device done at %.2f ms" % ((ts[-1] - ts[0]) * 1e3, (te - ts[0]) * 1e3))
