import sys, time, torch
sys.path.insert(0, '.')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx, chp, vhp = bench.build_context(0)
eng, chunks = bench.make_engine(ctx, B, 0)
st, slots, seg = eng.st, eng.slots, eng.seg
codes = torch.randint(0, 100, (B, seg), dtype=torch.int32, device='cuda')
for _ in range(5): st.decoder_step(slots, codes)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): st.decoder_step(slots, codes)
torch.cuda.synchronize(); print("decoder step alone %.3f ms" % ((time.perf_counter() - t0) / 40 * 1e3))
