"""Developer probe: each stage of the chunk step alone at 64 streams (back-to-back launches on one stream), against the
pipelined full step - what the stages cost each other when they share the chip."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx, chp, vhp = bench.build_context(0)
eng, chunks = bench.make_engine(ctx, B, 0)
st, slots, seg, hop = eng.st, eng.slots, eng.seg, ctx.hop
mel = torch.randn(B, seg, 80, device='cuda') * 0.5
codes = torch.randint(0, 100, (B, seg), dtype=torch.int32, device='cuda')
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
j = [0]
def emf():
    st.emformer_step(slots, chunks[j[0] % len(chunks)]); j[0] += 1
print("emformer step alone  %.3f ms" % timeit(emf))
print("decoder step alone   %.3f ms" % timeit(lambda: st.decoder_step(slots, codes)))
print("vocoder step alone   %.3f ms" % timeit(lambda: st.hifigan_step(slots, mel)))
bufs = [torch.empty(B, seg * hop, device='cuda') for _ in range(4)]
cd = torch.empty(B, seg, dtype=torch.int32, device='cuda'); mo = torch.empty(B, seg, 80, device='cuda')
k = [0]
def full():
    st.step_async(slots, chunks[k[0] % len(chunks)], bufs[k[0] % 4], emit=seg, codes=cd, mel_out=mo); k[0] += 1
t = timeit(full, 60); st.join(); torch.cuda.synchronize()
print("pipelined full step  %.3f ms" % t)
