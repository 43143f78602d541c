"""Developer probe: intervals between step completions from a cold start (python3 tools/cold_series.py [idle seconds])."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
ctx, chp, vhp = bench.build_context(0)
run = bench.Runner(ctx, bench.WORKLOADS["b64"], 64, 0, 1, "auto")
st = run.eng.st
spin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0      # seconds of unrelated GPU work (a GEMM loop) between the idle period and the steps
A = torch.randn(4096, 4096, device="cuda"); Bm = torch.randn(4096, 4096, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); time.sleep(idle)
    if spin > 0 and rep >= 1:
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < spin:
            for _ in range(10): A @ Bm
            torch.cuda.synchronize()
    N = 120
    st.step_clock(N + 1)
    t0 = time.perf_counter()
    for _ in range(N + 1):
        run.step()
    th = time.perf_counter() - t0
    run.barrier()
    iv = st.step_clock_read()
    st.step_clock(0)
    print("rep %d: host enqueue of %d steps %.1f ms; intervals (ms), means of 5:" % (rep, N + 1, th * 1e3))
    print("  " + " ".join("%.3f" % (sum(iv[i:i + 5]) / 5) for i in range(0, len(iv) - 4, 5)))
