"""Developer probe: stage timeline of pipelined steps at B streams (python3 tools/timeline.py [B] [steps]).
Prints, averaged over the steps after a warm-up: the duration of each stage, how long the vocoder stream sat idle between
the end of step t-1 and the start of step t, and how late the decoder of step t finished relative to the end of vocoder t-1
(positive = the vocoder had to wait for its mel)."""
import sys, statistics as st_
import torch
sys.path.insert(0, '.')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
WL = bench.WORKLOADS[sys.argv[3]] if len(sys.argv) > 3 else bench.WORKLOADS["b64"]      # optional third argument: a stateful bench workload (chunk size, memory bank)
ctx, chp, vhp = bench.build_context(0, WL["chunk_ms"], WL.get("memory", 0))
eng, chunks = bench.make_engine(ctx, B, 0)
st, slots, seg, hop = eng.st, eng.slots, eng.seg, ctx.hop
bufs = [torch.empty(B, seg * hop, device='cuda') for _ in range(4)]
cd = torch.empty(B, seg, dtype=torch.int32, device='cuda'); mo = torch.empty(B, seg, 80, device='cuda')
def step(k): st.step_async(slots, chunks[k % len(chunks)], bufs[k % 4], emit=seg, codes=cd, mel_out=mo)
for k in range(10): step(k)
st.join(); torch.cuda.synchronize()
st.step_timeline(N)
for k in range(N): step(10 + k)
st.join(); torch.cuda.synchronize()
tl = st.step_timeline_read(N)
st.step_timeline(0)
w = 8
dur = lambda a, b: st_.fmean(t[b] - t[a] for t in tl[w:])
print("steps %d: step interval %.3f ms (voc end to voc end)" % (len(tl), (tl[-1][5] - tl[w][5]) / (len(tl) - 1 - w)))
print("emformer stage %.3f ms, decoder stage %.3f ms, vocoder stage %.3f ms" % (dur(0, 1), dur(2, 3), dur(4, 5)))
print("vocoder stream idle between steps %.3f ms" % st_.fmean(tl[i][4] - tl[i - 1][5] for i in range(w, len(tl))))
print("decoder(t) end minus vocoder(t-1) end %.3f ms" % st_.fmean(tl[i][3] - tl[i - 1][5] for i in range(w, len(tl))))
print("decoder(t) start minus emformer(t) end %.3f ms; emformer(t) start minus emformer(t-1) end %.3f ms" % (
    st_.fmean(tl[i][2] - tl[i][1] for i in range(w, len(tl))), st_.fmean(tl[i][0] - tl[i - 1][1] for i in range(w, len(tl)))))
for t in tl[w:w + 4]: print("  ", " ".join("%8.3f" % (x - tl[w][0]) for x in t))
