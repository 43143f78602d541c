import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
ctx, chp, vhp = bench.build_context(0)
for B in (64, 1):
    eng, chunks = bench.make_engine(ctx, B, 0)
    seg, hop = eng.seg, ctx.hop
    codes = torch.empty(B, seg, dtype=torch.int32, device='cuda'); mel = torch.empty(B, seg, 80, device='cuda'); wav = torch.empty(B, seg*hop, device='cuda')
    ch = chunks[0].clone()
    def step(): eng.st.step(eng.slots, ch, emit=seg, codes=codes, mel_out=mel, wav_out=wav)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 30
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 30
    print(f"B={B}: eager {te*1e3:.3f} ms/step, graph replay {tg*1e3:.3f} ms/step")
    eng.st.close()
