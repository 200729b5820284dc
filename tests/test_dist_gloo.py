"""world_size-2 `gloo` tests (CPU) of the batch-data-parallel helpers: shard arithmetic, parameter
broadcast and the flat-bucket gradient all-reduce on real TT modules (strided core Parameters)."""
import contextlib
import io
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmpdir):
    for p in (ROOT, os.path.join(ROOT, "tensorized-rnn_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tensorized_rnn.tt_lstm import TTLSTM
        from ttrnn_hip.dist import FlatGradAllReduce, broadcast_parameters, shard_batch
        torch.manual_seed(100 + rank)            # different init per rank on purpose
        with contextlib.redirect_stdout(io.StringIO()):
            m = TTLSTM(28, 64, 2, torch.device("cpu"), n_cores=2, tt_rank=3)
        broadcast_parameters(m, src=0)
        # every rank now holds rank 0's parameters, strides untouched
        flat = torch.cat([p.detach().contiguous().view(-1) for p in m.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        assert any(not p.is_contiguous() for p in m.parameters())

        # fabricate rank-dependent gradients with the parameters' own (transposed) layout
        g = torch.Generator().manual_seed(7)
        base = [torch.randn(p.shape, generator=g) for p in m.parameters()]
        params = list(m.parameters())
        for i, (p, b) in enumerate(zip(params, base)):
            if i == 1 and rank == 1:
                p.grad = None                     # a rank may have no grad for a tensor: counts as zero
                continue
            gr = torch.empty_strided(p.shape, p.stride())
            gr.copy_(b * (rank + 1))
            p.grad = gr
        red = FlatGradAllReduce(m)
        assert red.nbytes == 4 * sum(p.numel() for p in params)
        red.sync()
        for i, (p, b) in enumerate(zip(params, base)):
            expect = b * (1.0 if i == 1 else 1.5)        # mean of 1x and 2x (or 1x and 0)
            if i == 1:
                expect = b * 0.5
            assert torch.allclose(p.grad, expect, atol=1e-6), i
            assert p.grad.stride() == p.stride()

        # batch sharding: contiguous, covers everything, no overlap
        x = torch.arange(7 * 3).view(7, 3)
        mine = shard_batch(x)
        parts = [torch.zeros(4, 3, dtype=x.dtype) for _ in range(world)]
        pad = torch.zeros(4, 3, dtype=x.dtype)
        pad[:mine.shape[0]] = mine
        dist.all_gather(parts, pad)
        sizes = [4, 3]
        assert torch.equal(torch.cat([parts[r][:sizes[r]] for r in range(world)]), x)
        with open(os.path.join(tmpdir, "ok%d" % rank), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_grad_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def test_shard_bounds():
    sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
    from ttrnn_hip.dist import shard_bounds
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["forward", "train"])
def test_bench_two_ranks_on_one_device(mode):
    """The N > 1 path of bench.py exactly as the driver launches it (torch.distributed.run, one rank per process, barrier +
    max-over-ranks timing, rank 0 prints ONE JSON line) on a one-GPU box: both ranks share cuda:0
    (TTRNN_BENCH_SINGLE_DEVICE=1) and rendezvous over gloo, because RCCL refuses two ranks on one device.  Checks the JSON
    contract of both modes; says nothing about scaling."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TTRNN_BENCH_SINGLE_DEVICE="1", TTRNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + os.getpid() % 200 + (7 if mode == "train" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--mode", mode]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["scaling"] == "weak" and rec["higher_is_better"] is True and rec["unit"] == "timesteps/s"
    assert rec["value"] > 0 and rec["ms_per_step"] > 0
    assert abs(rec["value"] - 2 * 784 / (rec["ms_per_step"] * 1e-3)) <= 1e-6 * rec["value"]      # whole-job aggregate
    assert rec["config"]["global_batch"] == 128 and "cpu_baseline" not in rec
    assert rec["roofline"]["frac"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_one_rank_group_drives_every_collective():
    """RCCL itself, on the one GPU of the box: a world_size-1 `nccl` process group through which the parameter broadcast,
    the flat-bucket gradient all-reduce (forced past its one-rank early-out), the GE2E all-gather and a barrier run on
    device tensors (tests/_rccl_one_rank.py, a fresh process).  Says nothing about scaling; it makes the first multi-GPU run
    not be the first RCCL run."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, universal_newlines=True, timeout=540, env=env, cwd=ROOT)
    assert res.returncode == 0 and "RCCL_ONE_RANK_OK" in res.stdout, (res.stdout[-2000:], res.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["forward", "train"])
def test_bench_distributed_branch_on_rccl_with_one_rank(mode):
    """`bench.py --gpus 1` with TTRNN_BENCH_FORCE_DIST=1: the code path of `--gpus N` (process group on the nccl backend with
    device_id, barriers, MAX all-reduce of the elapsed time, the flat-bucket gradient all-reduce of the train step) on RCCL
    with one rank."""
    import json
    import subprocess
    env = dict(os.environ, TTRNN_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29900 + os.getpid() % 90 + (5 if mode == "train" else 0)))
    env.pop("TTRNN_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--mode", mode,
           "--no-cpu-baseline"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=800, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["collectives"].startswith("nccl process group, world 1")
    assert ("gradient all-reduce" in rec["collectives"]) == (mode == "train")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_strong_scaling_two_ranks_on_one_device():
    """`--scaling strong`: the configuration's batch is the GLOBAL batch; two ranks (gloo, sharing cuda:0)
    take 32 of cfg2's 64 samples each, and `value` counts the global batch's timesteps once."""
    import json
    import subprocess
    env = dict(os.environ, TTRNN_BENCH_SINGLE_DEVICE="1", TTRNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--scaling", "strong"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=800, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2
    assert rec["config"]["global_batch"] == 64 and rec["config"]["per_gpu_batch"] == 32
    assert abs(rec["value"] - 784 / (rec["ms_per_step"] * 1e-3)) <= 1e-6 * rec["value"]
    assert abs(rec["sample_timesteps_per_s"] - 64 * rec["value"]) <= 1e-6 * rec["sample_timesteps_per_s"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_shard_of_times_one_ranks_strong_scaling_shard():
    """`bench.py --shard-of 4` (single-GPU diagnostic): rank 0's slice of cfg2's 64 samples under a 4-GPU strong-scaling job —
    16 samples, labelled `shard_of`, counted as one batch advancing T timesteps per step; refused together with a real
    multi-rank launch flag."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--shard-of", "4", "--no-cpu-baseline"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=500, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["config"]["per_gpu_batch"] == 16 and rec["config"]["global_batch"] == 16 and rec["n_gpus"] == 1
    assert "4 GPUs" in rec["shard_of"] and "not a scaling measurement" in rec["shard_of"]
    assert abs(rec["value"] - 784 / (rec["ms_per_step"] * 1e-3)) <= 1e-6 * rec["value"]
    bad = subprocess.run(cmd + ["--scaling", "strong"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                         timeout=500, env=env, cwd=ROOT)
    assert bad.returncode != 0 and "single-GPU diagnostic" in bad.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,scaling", [("forward", "weak"), ("train", "weak"), ("forward", "strong"), ("train", "strong")])
def test_bench_plain_invocation_launches_its_own_ranks(mode, scaling):
    """Plain `python bench.py --gpus 2` (no torchrun environment): the process becomes the launcher, starts
    `python -m torch.distributed.run --nproc-per-node 2 bench.py ...` as a CHILD, prints rank 0's JSON line last and exits
    with the child's code.  Two gloo ranks share cuda:0 here (TTRNN_BENCH_SINGLE_DEVICE=1); `ranks_seen` = an all-reduce of
    ones over the group."""
    import json
    import subprocess
    env = dict(os.environ, TTRNN_BENCH_SINGLE_DEVICE="1", TTRNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", mode,
           "--scaling", scaling]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=800, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    out_lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert out_lines[-1].startswith('{"metric"'), out_lines[-3:]
    assert sum(1 for ln in out_lines if ln.startswith('{"metric"')) == 1
    rec = json.loads(out_lines[-1])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["scaling"] == scaling
    assert rec["checked"]["finite"] is True and rec["checked"]["ranks_bad"] == 0
    assert rec["checked"]["device_status"]["pair_timeouts"] == 0
    assert rec["checked"]["routes"]["forward"] == "fused_core"
    assert (rec["checked"]["routes"]["backward"] is not None) == (mode == "train")
    assert rec["config"]["global_batch"] == (128 if scaling == "weak" else 64)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_reports_and_fails_on_a_pair_timeout():
    """The bench line vouches for what it timed: with the pair kernels' time-out path planted (option pair_fault, the
    H = 1024 class at 2B <= #CUs: `--shard-of 8` of cfg5 = 16 samples) the affected samples are NaN by construction —
    the line says so (`checked.finite` false, `device_status.pair_timeouts` > 0) and the exit code is non-zero.  The same
    command without the fault exits 0 with a clean line."""
    import json
    import subprocess
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        base.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg5", "--shard-of", "8", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline"]
    ok = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=800, env=base, cwd=ROOT)
    assert ok.returncode == 0, ok.stderr[-3000:]
    rec = json.loads([ln for ln in ok.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert rec["checked"]["finite"] is True and rec["checked"]["device_status"]["pair_timeouts"] == 0
    assert rec["checked"]["routes"]["forward"] == "merged_big"
    assert rec["dtype"].startswith("f32 (2xfp16 operands")
    bad = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=800,
                         env=dict(base, TTRNN_PAIR_FAULT="1"), cwd=ROOT)
    assert bad.returncode != 0, bad.stdout[-1500:]
    rec = json.loads([ln for ln in bad.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert rec["checked"]["device_status"]["pair_timeouts"] > 0 and rec["checked"]["ranks_bad"] == 1
    assert "not valid" in bad.stderr


def test_bench_launcher_passes_through_and_never_touches_the_gpu(tmp_path):
    """The launcher half of `python bench.py --gpus N` on a box without a GPU: it must start N child ranks (which then
    refuse to run — no GPU here) and hand their exit code back, without itself calling into torch.cuda or libttrnn.  Run
    with a poisoned `torch.cuda.is_available` in a sitecustomize that arms itself in the PARENT only (argv[0] is bench.py
    and there is no WORLD_SIZE) — a parent that probes the GPU fails the test."""
    import subprocess
    site = tmp_path / "sitecustomize.py"
    site.write_text(
        "import os, sys\n"
        "if 'WORLD_SIZE' not in os.environ and sys.argv and sys.argv[0].endswith('bench.py'):\n"
        "    import torch\n"
        "    def _boom(*a, **k):\n"
        "        raise SystemExit('LAUNCHER_TOUCHED_THE_GPU')\n"
        "    torch.cuda.is_available = _boom\n"
        "    torch.cuda.set_device = _boom\n"
        "    torch.cuda.current_device = _boom\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300, env=env, cwd=ROOT)
    assert "LAUNCHER_TOUCHED_THE_GPU" not in res.stderr + res.stdout
    if not __import__("torch").cuda.is_available():
        # the ranks were started (torchrun ran them) and each refused: that message can only come from a child with WORLD_SIZE=2
        assert res.returncode != 0
        assert "bench.py needs a GPU" in res.stderr, res.stderr[-2000:]
