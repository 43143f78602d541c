import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
ctx, chp, vhp = bench.build_context(0)
for parts in (1, 2, 4):
    B = 64 // parts
    engs = [bench.make_engine(ctx, B, first_stream=k * B) for k in range(parts)]
    seg, hop = engs[0][0].seg, ctx.hop
    bufs = [[torch.empty(B, seg * hop, device='cuda') for _ in range(4)] for _ in range(parts)]
    codes = [torch.empty(B, seg, dtype=torch.int32, device='cuda') for _ in range(parts)]
    mels = [torch.empty(B, seg, 80, device='cuda') for _ in range(parts)]
    def step(j):
        for k, (eng, chunks) in enumerate(engs):
            eng.st.step_async(eng.slots, chunks[j % len(chunks)], bufs[k][j % 4], emit=seg, codes=codes[k], mel_out=mels[k])
    def sync():
        for eng, _ in engs: eng.st.join()
        torch.cuda.synchronize()
    for j in range(8): step(j)
    sync()
    t0 = time.perf_counter()
    N = 40
    for j in range(8, 8 + N): step(j)
    sync()
    dt = (time.perf_counter() - t0) / N
    print(f"parts={parts} (B={B} each): {dt*1e3:.3f} ms per 64-stream step = {64/dt:.0f} chunks/s")
    for eng, _ in engs: eng.st.close()
