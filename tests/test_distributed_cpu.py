"""world_size-2 gloo test of the only exchange on the path: gather of finished audio to rank 0
(stream-sharded data parallelism)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from conan_amd.engine import gather_audio, gather_audio_equal, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    # each stream's "audio" is its global stream id, so the gathered tensor must come back in stream order
    wav = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 8)
    out = gather_audio(wav, world, rank)
    ok = True
    if rank == 0:
        ok = out.shape == (total, 8) and torch.equal(out[:, 0], torch.arange(total, dtype=torch.float32))
    else:
        ok = out is None
    if (hi - lo) * world == total:       # equal shards: the benchmark's fast path
        bufs = gather_audio_equal(wav, world, rank)
        if rank == 0:
            ok = ok and torch.equal(torch.cat(bufs)[:, 0], torch.arange(total, dtype=torch.float32))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("total", [8, 5])
def test_gather_audio_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _ring_worker(rank, world, port, steps, q, every=1):
    """The benchmark's step / gather choreography (bench.py: AudioGatherRing) on a gloo group with a stub engine: step j
    of rank r "computes" audio filled with 1000 r + j into the ring buffer it was handed; rank 0 must receive every
    rank's audio of every step, in step order, although buffers are reused every `nb` steps."""
    from conan_amd.engine import AudioGatherRing
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []

    def on_gathered(j, bufs):
        seen.append((j, [float(b[0, 0]) for b in bufs], all(bool((b == b[0, 0]).all()) for b in bufs)))

    ring = AudioGatherRing(lambda: torch.zeros(3, 16), world, rank, nb=4, on_gathered=on_gathered, every=every)
    joined = []
    for j in range(steps):
        buf, fence = ring.acquire(j, fence=True)      # the benchmark's form: the step takes the side stream as its output fence
        assert fence is None                          # (host-only group: everything is synchronous, nothing to wait for)
        buf.fill_(1000.0 * rank + j)                  # the "vocoder" of step j
        ring.submit(j, join=lambda j=j: joined.append(j))
    ring.flush(steps - 1)                             # a run that ends inside a group: the rest travels now
    ring.drain()
    dist.barrier()
    dist.destroy_process_group()
    # one join + one collective per `every` steps (+ the flush of the incomplete last group)
    want_gathers = (steps + every - 1) // every
    want_joined = [j for j in range(steps) if (j + 1) % every == 0] + ([steps - 1] if steps % every else [])
    ok = joined == want_joined and ring.submitted == want_gathers and ring.last_gathered == steps - 1
    if rank == 0:
        ok = ok and [s[0] for s in seen] == list(range(steps))
        ok = ok and all(vals == [1000.0 * r + j for r in range(world)] and uniform for j, vals, uniform in seen)
    else:
        ok = ok and seen == []
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world,every", [(2, 1), (2, 4), (8, 4)])
def test_bench_gather_choreography(world, every):
    """world 8 = the 8-GPU node's rank count (BASELINE.json configs[3]): the choreography the driver's N = 8 run executes
    (every = 4: one collective per four steps, bench.py's default; 11 steps end inside a group)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, 11, q, every)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def _midflush_worker(rank, world, port, q):
    """A flush() in the MIDDLE of a group, the run then continues: the flush must send only the steps that have
    run - the group's later buffers may be written while the collective reads - and the group's remainder must travel without
    the steps the flush already delivered.  Step j of rank r writes 1000 r + j; the buffers of steps that have not run hold -1."""
    from conan_amd.engine import AudioGatherRing
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen, sizes = [], []

    def on_gathered(j, bufs):
        seen.append((j, [float(b[0, 0]) for b in bufs]))

    ring = AudioGatherRing(lambda: torch.zeros(3, 16), world, rank, on_gathered=on_gathered, every=4)
    ring.pool.fill_(-1.0)
    steps = 14
    for j in range(steps):
        buf, fence = ring.acquire(j, fence=True)
        assert fence is None
        buf.fill_(1000.0 * rank + j)
        ring.submit(j)
        if j in (1, 9):                       # mid-group flushes (steps 0-1 and 8-9 travel early)
            ring.flush(j)
            sizes.append(tuple(ring.last_sent.shape))
            assert float(ring.last_sent.min()) >= 0.0      # nothing of a step that has not run
    ring.flush(steps - 1)
    ring.drain()
    dist.barrier()
    dist.destroy_process_group()
    ok = sizes == [(2, 3, 16), (2, 3, 16)] and ring.submitted == 6 and ring.last_gathered == steps - 1
    if rank == 0:
        ok = ok and [s[0] for s in seen] == list(range(steps)) and all(v == [1000.0 * r + j for r in range(world)] for j, v in seen)
    q.put((rank, bool(ok)))


def test_gather_ring_flush_inside_a_group_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_midflush_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_gather_ring_fence_only_for_recorded_groups():
    """acquire(fence=True) must not hand out the event of a group whose gather never ran (submit() skipped): an unrecorded
    event is a null handle for conan_streams_output_fence_event."""
    from conan_amd.engine import AudioGatherRing
    ring = AudioGatherRing(lambda: torch.zeros(2, 8), 1, 0, always=True, every=4)
    for j in range(12):
        buf, fence = ring.acquire(j, fence=True)          # no submit() at all
        assert fence is None
    assert ring.recorded == [False, False]


def _check_worker(rank, world, port, q):
    """bench.py's rank bookkeeping on a host group: per-rank clocks and devices by all_gather, and the check that what
    rank 0 gathered for the last step is what every rank produced (integer checksums of the fp32 bit patterns)."""
    from conan_amd.engine import AudioGatherRing
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ring = AudioGatherRing(lambda: torch.zeros(4, 32), world, rank, nb=4, every=4)
    g = torch.Generator().manual_seed(100 + rank)
    steps = 6
    for j in range(steps):
        buf, _ = ring.acquire(j, fence=True)
        buf.copy_(torch.randn(4, 32, generator=g))
        ring.submit(j)
    ring.flush(steps - 1)
    ring.drain()
    assert ring.last_gathered == steps - 1 and ring.submitted == 2
    last = ring.last_sent                      # the group of step buffers this rank handed to the last gather
    csum = last.view(torch.int32).to(torch.int64).sum().reshape(1)
    sums = [torch.zeros_like(csum) for _ in range(world)]
    dist.all_gather(sums, csum)
    mine = torch.tensor([0.001 * (rank + 1), float(rank)], dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    ok = [int(r[1].item()) for r in allr] == list(range(world)) and max(float(r[0]) for r in allr) == pytest.approx(0.001 * world)
    if rank == 0:
        got = [int(b.view(torch.int32).to(torch.int64).sum().item()) for b in ring.gbufs]
        ok = ok and got == [int(x.item()) for x in sums] and len(set(got)) == world
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_bench_gather_check_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_check_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


# ---- bench.py starts its own ranks (task 1): `python bench.py --gpus N` without a launcher must measure N GPUs ----

def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launcher_command_and_no_recursion():
    b = _bench_module()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = b.launcher_command(8, argv, env={}, port=29611)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    k = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[k + 1:] == argv                                  # the ranks get the very same arguments
    # a process that is a rank already (started by torch.distributed.run, the driver's N > 1 form) never spawns again
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        assert b.launcher_command(8, argv, env={var: "1"}) is None
    assert b.launcher_command(1, ["--gpus", "1"], env={}) is None
    # a free port is chosen when none is given
    assert int(b.launcher_command(2, [], env={})[cmd.index("--master-port") + 1]) > 0


def test_bench_rank_count_mismatch_is_an_error_not_a_warning():
    b = _bench_module()
    assert b.check_ranks(8, 8, 8) is None and b.check_ranks(1, 1, 1) is None
    assert "WORLD_SIZE=2" in b.check_ranks(8, 2, 8)
    assert "one process per GPU" in b.check_ranks(2, 2, 1)


def test_bench_asked_for_more_gpus_than_the_node_has_exits_nonzero():
    """This container has no GPU (a 1-GPU box asked for 2 takes the same branch): nothing is launched, exit code != 0."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "GPU(s)" in r.stderr
    assert r.stdout.strip() == ""                              # no JSON line for a job that did not run
    # a rank whose WORLD_SIZE disagrees with --gpus refuses as well
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
