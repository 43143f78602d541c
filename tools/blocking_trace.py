"""Developer probe: blocking fused steps at B streams for a kernel trace (rocprofv3 --kernel-trace -- python3 tools/blocking_trace.py [B] [auto|f32|limb])."""
import sys, torch
sys.path.insert(0, '.')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ARITH = sys.argv[2] if len(sys.argv) > 2 else 'auto'
ctx, chp, vhp = bench.build_context(0)
eng, chunks = bench.make_engine(ctx, B, 0, arith=ARITH)
seg, hop = eng.seg, ctx.hop
codes = torch.empty(B, seg, dtype=torch.int32, device='cuda'); mel = torch.empty(B, seg, 80, device='cuda'); wav = torch.empty(B, seg * hop, device='cuda')
for j in range(6):
    eng.st.step(eng.slots, chunks[j], emit=seg, codes=codes, mel_out=mel, wav_out=wav)
torch.cuda.synchronize()
eng.st.profile_mark()
for j in range(6, 12):
    eng.st.step(eng.slots, chunks[j], emit=seg, codes=codes, mel_out=mel, wav_out=wav)
eng.st.profile_mark()
torch.cuda.synchronize()
