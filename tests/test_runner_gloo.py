"""CPU, world_size 2 over gloo: pair sharding + the single metrics all_gather of the multi-GPU path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from nopesac_amd import runner
    r, w, _ = runner.init_distributed("gloo")
    lo, hi = runner.shard_range(n_pairs, r, w)
    B = hi - lo
    idx = torch.arange(lo, hi, dtype=torch.float32)
    trans = idx.view(-1, 1) * torch.tensor([1.0, 2.0, 3.0])
    rot = torch.nn.functional.normalize(torch.ones(B, 4) + idx.view(-1, 1), dim=-1)
    rows = runner.metric_rows(trans, rot, torch.full((B,), 5), torch.full((B,), 7), torch.full((B,), 3), lo,
                              t_err=idx * 0.1, r_err=idx * 2.0)
    allrows = runner.gather_metrics(rows)
    torch.distributed.barrier()
    q.put((rank, allrows.numpy(), runner.summarize(allrows)))
    torch.distributed.destroy_process_group()


def test_shard_and_gather_world2():
    n_pairs, world = 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rows, summ in got:
        assert rows.shape == (n_pairs, 16)
        assert rows[:, 12].tolist() == list(range(n_pairs))            # contiguous shards, rank order
        np.testing.assert_allclose(rows[:, 0], np.arange(n_pairs))
        assert summ["pairs"] == n_pairs and abs(summ["mean_T_err"] - 0.35) < 1e-6 and summ["mean_matches"] == 3.0
    np.testing.assert_array_equal(got[0][1], got[1][1])


def test_shard_range_covers_everything():
    from nopesac_amd import runner
    for n, w in ((256, 8), (10, 4), (3, 8), (1, 1)):
        seen = []
        for r in range(w):
            lo, hi = runner.shard_range(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))


def test_pose_error_formulas():
    from nopesac_amd import runner
    q = np.array([[1.0, 0, 0, 0], [np.cos(0.25), np.sin(0.25), 0, 0]])
    assert np.allclose(runner.rotation_error_deg(q, q), 0, atol=1e-3)
    assert np.allclose(runner.rotation_error_deg(q[:1], q[1:]), np.degrees(0.5), atol=1e-4)
    assert np.allclose(runner.rotation_error_deg(q[:1], -q[1:]), np.degrees(0.5), atol=1e-4)   # sign invariant
    assert np.allclose(runner.translation_error(np.array([[3.0, 4, 0]]), np.zeros((1, 3))), 5.0)


def _launched_main(n_pairs, out_dir):
    """What run.py's per-rank main does, minus the GPU model: join the group, take the shard, gather the rows."""
    from nopesac_amd import runner
    r, w, local = runner.init_distributed("gloo")
    lo, hi = runner.shard_range(n_pairs, r, w)
    idx = torch.arange(lo, hi, dtype=torch.float32)
    B = hi - lo
    rows = runner.metric_rows(idx.view(-1, 1).repeat(1, 3), torch.ones(B, 4) / 2, torch.ones(B), torch.ones(B), torch.zeros(B), lo)
    allrows = runner.gather_metrics(rows)
    np.save(os.path.join(out_dir, f"rows_{r}.npy"), allrows.numpy())
    torch.distributed.destroy_process_group()


def test_launch_spawns_one_rank_per_gpu(tmp_path, monkeypatch):
    """`run.py --num-gpus N` without torchrun: runner.launch starts the ranks itself (detectron2 `launch`,
    test_NopeSAC.py:209-216)."""
    from nopesac_amd import runner
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert runner.launch(lambda a: a + 1, 1, (1,)) == 2                    # single GPU: plain call
    runner.launch(_launched_main, 2, (6, str(tmp_path)))
    a, b = np.load(tmp_path / "rows_0.npy"), np.load(tmp_path / "rows_1.npy")
    assert a.shape == (6, 16) and a[:, 12].tolist() == list(range(6))
    np.testing.assert_array_equal(a, b)


def _inflight_worker(rank, world, port, steps, slots, B, out_dir):
    """bench.py's timed loop (runner.InflightLoop: `slots` batches in flight, ONE gather_metrics per step) with a stub model."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from nopesac_amd import runner
    r, w, _ = runner.init_distributed("gloo")
    loop = runner.InflightLoop(slots, B, None, w)
    seen = []

    def device_step_for(i):
        def device_step(slot):
            assert slot == i % slots
            t = torch.full((B, 3), float(i)) + torch.arange(B, dtype=torch.float32).view(-1, 1) / 100 + r * 1000.0
            q = torch.nn.functional.normalize(torch.ones(B, 4), dim=-1)
            rows = runner.metric_rows(t, q, torch.full((B,), 5), torch.full((B,), 6), torch.full((B,), 32), r * B)
            return {"step": i}, rows
        return device_step

    for i in range(steps):
        d, host = loop.step(i, device_step_for(i))
        assert d == {"step": i} and host.shape == (w * B, runner.METRIC_WIDTH)
        seen.append(host.clone())                         # (a slot's buffer is overwritten `slots` steps later)
    loop.barrier()
    assert loop.host_seconds > 0 and loop.last == ({"step": steps - 1}, (steps - 1) % slots)
    np.save(os.path.join(out_dir, f"inflight_{r}.npy"), torch.stack(seen).numpy())
    torch.distributed.destroy_process_group()


def test_bench_inflight_loop_world2(tmp_path):
    """The multi-GPU form of bench.py's timed region on CPU: world size 2 over gloo, 4 slots, 9 steps (more steps than slots: every
    slot is reused), one all_gather per step and rank.  Both ranks must see identical [world*B, 16] blocks for every step, rank-major
    with contiguous pair indices, and the loop must terminate (no deadlock from the per-slot ordering of the collectives)."""
    steps, slots, B, world = 9, 4, 3, 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_inflight_worker, args=(r, world, port, steps, slots, B, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, "rank did not finish (deadlock?)"
    a, b = np.load(tmp_path / "inflight_0.npy"), np.load(tmp_path / "inflight_1.npy")
    assert a.shape == (steps, world * B, 16)
    np.testing.assert_array_equal(a, b)
    for i in range(steps):
        assert a[i, :, 12].tolist() == list(range(world * B))                      # rank-major, contiguous pair indices
        np.testing.assert_allclose(a[i, :B, 0], i + np.arange(B) / 100, rtol=1e-6)           # rank 0's rows of step i
        np.testing.assert_allclose(a[i, B:, 0], 1000 + i + np.arange(B) / 100, rtol=1e-6)    # rank 1's rows of step i
        assert (a[i, :, 9] == 32).all()


def _grouped_worker(rank, world, port, steps, slots, B, G, out_dir):
    """InflightLoop(gather_every = G): the rows of G steps per all_gather, the last partial group inside barrier()."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from nopesac_amd import runner
    r, w, _ = runner.init_distributed("gloo")
    loop = runner.InflightLoop(slots, B, None, w, gather_every=G)
    groups, seen = [], 0

    def device_step_for(i):
        def device_step(slot):
            t = torch.full((B, 3), float(i)) + torch.arange(B, dtype=torch.float32).view(-1, 1) / 100 + r * 1000.0
            q = torch.nn.functional.normalize(torch.ones(B, 4), dim=-1)
            return {"step": i}, runner.metric_rows(t, q, torch.full((B,), 5), torch.full((B,), 6), torch.full((B,), 32), r * B)
        return device_step

    for i in range(steps):
        d, host = loop.step(i, device_step_for(i))
        assert d == {"step": i} and host is None
        if loop.collectives > seen:                       # this step completed a group
            seen = loop.collectives
            groups.append(loop.last_group_rows().clone())
    loop.barrier()                                        # flush: the partial group
    if loop.collectives > seen:
        groups.append(loop.last_group_rows().clone())
    assert loop.collectives == -(-steps // G)
    last = loop.last_step_rows()
    assert last.shape == (w * B, runner.METRIC_WIDTH) and last[:, 12].tolist() == [float(v) for v in range(w * B)]
    np.save(os.path.join(out_dir, f"grouped_{r}.npy"), torch.cat(groups, dim=1).numpy())       # [world, steps, B, 16]
    torch.distributed.destroy_process_group()


def test_inflight_loop_gather_every_world2(tmp_path):
    """SURVEY 8(e) / round-5 hardening: ranks are coupled once per G steps, not every step.  World 2 over gloo, 4 slots, 11 steps, G = 4:
    three collectives (4 + 4 + 3 steps), every step's rows of both ranks arrive, in step order and rank-major, identical on both ranks."""
    steps, slots, B, world, G = 11, 4, 3, 2, 4
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_grouped_worker, args=(r, world, port, steps, slots, B, G, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, "rank did not finish (deadlock?)"
    a, b = np.load(tmp_path / "grouped_0.npy"), np.load(tmp_path / "grouped_1.npy")
    assert a.shape == (world, steps, B, 16)
    np.testing.assert_array_equal(a, b)
    for i in range(steps):
        for rk in range(world):
            np.testing.assert_allclose(a[rk, i, :, 0], 1000.0 * rk + i + np.arange(B) / 100, rtol=1e-6)
            assert a[rk, i, :, 12].tolist() == list(range(rk * B, rk * B + B))


def _run_bench(argv, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun environment must start two ranks by itself (runner.launch) - not run one rank and
    report n_gpus 1.  Stub model (fabricated rows) on CPU over gloo: the launcher, InflightLoop, the per-step all_gather, the timed
    region's max-over-ranks and rank 0's single JSON line are bench.py's own."""
    r, j = _run_bench(["--gpus", "2", "--stub-model", "--steps", "6", "--warmup", "2", "--pairs", "3", "--inflight", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1          # rank 0 only
    assert j["n_gpus"] == 2 and j["config"]["rccl_ranks"] == 2 and j["config"]["global_batch"] == 6
    assert j["rows_gathered"] == 6 and j["rows_in_rank_order"] and j["steps"] == 6 and j["warmup"] == 2
    assert j["INVALID_stub_model"] is True


def test_bench_gpus_mismatch_fails_loudly():
    """A torchrun world of another size than --gpus, or more GPUs asked for than visible, is an error - never a silent 1-rank run."""
    r, j = _run_bench(["--gpus", "4", "--stub-model", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and j is None and "WORLD_SIZE" in r.stderr
    r, j = _run_bench(["--gpus", "64", "--steps", "1", "--warmup", "0"])              # the real model path: no node has 64 GPUs
    assert r.returncode != 0 and j is None and "visible" in r.stderr


def test_bench_single_rank_stub():
    r, j = _run_bench(["--gpus", "1", "--stub-model", "--steps", "3", "--warmup", "1", "--pairs", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert j["n_gpus"] == 1 and j["config"]["rccl_ranks"] == 1 and j["rows_gathered"] == 4


def test_bench_gpus8_global_batch_256_stub():
    """BASELINE configs[3] in shape (8 ranks x 32 pairs = global batch 256, four steps in flight, one all_gather of [32,16] rows per
    step and rank) through bench.py's own launcher on CPU / gloo with the stub model: the 8-rank rendezvous, the rank-major order
    of the 256 gathered rows and the single JSON line are what a node without torchrun would run."""
    r, j = _run_bench(["--gpus", "8", "--stub-model", "--steps", "6", "--warmup", "2", "--pairs", "32", "--inflight", "4", "--no-tape"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert j["n_gpus"] == 8 and j["config"]["rccl_ranks"] == 8 and j["config"]["global_batch"] == 256
    assert j["rows_gathered"] == 256 and j["rows_in_rank_order"]
    assert j["steps_per_all_gather"] == 8 and j["collectives_in_run"] == 2          # warm-up flush (2 steps) + the timed region's 6 steps


@pytest.mark.parametrize("fail_rank", [None, "0", "1"])
def test_bench_tape_leg_world2_one_rank_fails_capture(fail_rank):
    """The launch-tape leg under world > 1 (round 6; single-process only before: a rank whose capture failed went back to eager
    launching while the others walked into the replay check's barrier).  bench.py's stub model drives the PROTOCOL
    (runner.capture_on_all_ranks: capture locally, all-reduce an ok flag, collective replay check, all-reduce again) over gloo at world
    2: with no failure both ranks replay and the timed region runs a second time; with the capture of rank 0 or of rank 1 forced to
    fail (NOPESAC_FAIL_CAPTURE_RANK) BOTH ranks stay eager, nothing hangs, the run ends with its single JSON line."""
    env = {} if fail_rank is None else {"NOPESAC_FAIL_CAPTURE_RANK": fail_rank}
    r, j = _run_bench(["--gpus", "2", "--stub-model", "--steps", "5", "--warmup", "2", "--pairs", "3", "--inflight", "4", "--gather-every", "2"], env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert j["n_gpus"] == 2 and j["rows_gathered"] == 6 and j["rows_in_rank_order"]
    t = j["launch_tape"]
    if fail_rank is None:
        assert t["captured_on_every_rank"] and t["replaying"] and t["checked_steps"] == 4 and t["ms_per_step"] is not None
    else:
        assert not t["captured_on_every_rank"] and not t["replaying"] and t["checked_steps"] == 0 and t["ms_per_step"] is None


def _capture_protocol_worker(rank, world, port, case, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from nopesac_amd import runner
    r, w, _ = runner.init_distributed("gloo")
    log = []
    calls = {"verify": 0, "abandon": 0}

    def capture_local():
        if case == "capture_raises_on_1" and r == 1:
            raise RuntimeError("no graph on this rank")

    def verify_collective():
        calls["verify"] += 1
        torch.distributed.barrier()                          # the check's own collectives: every rank must get here or nobody
        if case == "verify_raises_on_0" and r == 0:
            raise RuntimeError("replay differs")
        return not (case == "verify_false_on_1" and r == 1)

    def abandon():
        calls["abandon"] += 1
    ok = runner.capture_on_all_ranks(capture_local, verify_collective, abandon, None, log=log.append)
    torch.distributed.barrier()
    with open(os.path.join(out_dir, "cap_%s_%d.txt" % (case, r)), "w") as f:
        f.write("%d %d %d %d" % (int(ok), calls["verify"], calls["abandon"], len(log)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("case", ["all_ok", "capture_raises_on_1", "verify_raises_on_0", "verify_false_on_1"])
def test_capture_on_all_ranks_world2(tmp_path, case):
    """runner.capture_on_all_ranks at world 2 over gloo: every rank returns the SAME verdict; a local capture failure keeps every rank
    out of the collective check (verify not called anywhere); a failed check on one rank abandons on both; nothing deadlocks."""
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_capture_protocol_worker, args=(r, world, port, case, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = [tuple(int(v) for v in open(tmp_path / ("cap_%s_%d.txt" % (case, r))).read().split()) for r in range(world)]
    want = {"all_ok": (1, 1, 0, 0), "capture_raises_on_1": (0, 0, 1, 1), "verify_raises_on_0": (0, 1, 1, 1), "verify_false_on_1": (0, 1, 1, 1)}[case]
    assert got[0] == want and got[1] == want, got


def _run_cli(argv, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-m", "nopesac_amd.run"] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


def test_cli_world8_stub_matches_single_rank(tmp_path):
    """`python -m nopesac_amd.run --num-gpus 8` itself (not bench.py) at the world size of BASELINE configs[3], on CPU over gloo with the
    stub model: its own launcher, the contiguous shards (19 pairs over 8 ranks: 3 each, one rank with a single pair, one with NONE),
    the batch loop, the evaluator's padded all_gather and rank 0's summary must give exactly the single-rank run's tables; every rank
    is pinned to its share of the cores and sizes its decode pool from it (MODEL.AMD.CPU_AFFINITY).  Unmeasured on hardware."""
    import json
    common = ["--stub-model", "--synthetic-pairs", "19", "--pairs-per-batch", "2", "MODEL.DEVICE", "cpu"]
    r8 = _run_cli(["--num-gpus", "8", "--output", str(tmp_path / "w8.json")] + common)
    assert r8.returncode == 0, r8.stderr[-3000:]
    r1 = _run_cli(["--num-gpus", "1", "--output", str(tmp_path / "w1.json")] + common)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a, b = json.load(open(tmp_path / "w8.json")), json.load(open(tmp_path / "w1.json"))
    assert a["INVALID_stub_model"] and a["pairs"]["count"] == 19 == b["pairs"]["count"]
    for k in ("pairs", "camera", "camera_init", "camera_initRec", "camera_avgRef0", "camera_softRef0"):
        assert a[k].keys() == b[k].keys()
        for kk in a[k]:
            assert abs(a[k][kk] - b[k][kk]) < 1e-5, (k, kk, a[k][kk], b[k][kk])
    t = a["timing(rank0)"]
    assert t["pairs"] == 3 and t["pinned"] and t["decode_threads"] >= 1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if cores >= 8:
        assert t["cores_of_this_rank"] == cores // 8
    assert not b["timing(rank0)"]["pinned"]
