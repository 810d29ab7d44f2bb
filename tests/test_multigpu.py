"""Several GPUs, one process each, RCCL between them -- a test that ARMS ITSELF: world = min(8, visible GPUs), skipped
on a one-GPU box (where the same worker still runs at world size 1, so that its code cannot rot unseen).

tests/multigpu_worker.py under torch.distributed.run: every rank takes its byte range of S-single, S-wrapped, "tricky"
(a quality block of FASTQ-looking text: wrong entry guesses), "long" / "long-wrapped" (a record longer than the halo:
the look-ahead grows across ranks) and "small" streams, runs the library's own step (ffq_shard_*; ncclSend / ncclRecv
between different ranks, one ncclAllGather) plain, with the decode and pipelined over two lanes, then the file-backed
form (readfastq_iter_range over one shared file); rank 0 gathers every rank's rows and compares them with the oracle's
scan of the whole stream and with the run of the same ranges as k logical ranks in one process (same rows, same repair
rounds), asserts transport == "rccl" and handoff_bytes > 0.  The invariance being tested is the reference's own:
/root/reference/tests.py:219-226 (the entries do not depend on how the stream is cut into buffers)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from _ports import free_port as _free_port      # noqa: E402  (below the ephemeral range: tests/_ports.py)


def _launch(world, scratch, timeout, one_gpu=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if one_gpu:
        env["FFQ_TEST_RANKS_ON_ONE_GPU"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multigpu_worker.py"), str(scratch)]
    try:
        return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        return e


def n_gpus():
    try:
        return min(8, torch.cuda.device_count())
    except Exception:      # noqa: BLE001
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(n_gpus() < 2, reason="one GPU visible: RCCL between ranks needs at least two (the test arms itself on a multi-GPU box)")
def test_native_shards_over_rccl_all_visible_gpus(gpu_ctx, tmp_path):
    world = n_gpus()
    r = _launch(world, tmp_path, 1500)
    assert not isinstance(r, subprocess.TimeoutExpired), "multi-GPU worker timed out (world %d)" % world
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "multi-gpu shards ok: world %d" % world in r.stdout


@pytest.mark.gpu
def test_multigpu_worker_at_world_one(gpu_ctx, tmp_path):
    """The same worker under the same launcher with one rank: communicators, lanes, gathers, the file-backed ranges and
    rank 0's check all run (no peer to hand anything to)."""
    r = _launch(1, tmp_path, 900)
    assert not isinstance(r, subprocess.TimeoutExpired), "worker timed out"
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "multi-gpu shards ok: world 1" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world", (2, 3, 8))
def test_native_shards_over_rccl_with_real_peers_on_one_gpu(gpu_ctx, tmp_path, world):
    """RCCL WITH PEERS on a one-GPU box (round 6): `world` processes on GPU 0 that tell RCCL they sit on different hosts
    (NCCL_HOSTID per rank; tests/multigpu_worker.py), so that its duplicate-GPU check lets them into one communicator -- over the
    socket transport instead of xGMI, but the same librccl calls the product makes between GPUs: ncclCommInitRank of `world`
    ranks twice (hand-off and gather communicators), ncclSend / ncclRecv between DIFFERENT ranks in one group, the all-gather,
    both driven from their own streams at once (lanes), the serial one-communicator step, and the watchdog with a peer whose
    gather really never arrives: FFQ_E_TIMEOUT on every rank, ncclCommAbort on a collective that is stuck, a new communicator,
    the rows.  Everything the worker checks at world 1 and on a multi-GPU box, checked here against the oracle."""
    r = _launch(world, tmp_path, 1500, one_gpu=True)
    assert not isinstance(r, subprocess.TimeoutExpired), "worker timed out (world %d on one GPU)" % world
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-8000:]
    assert "multi-gpu shards ok: world %d" % world in r.stdout


@pytest.mark.gpu
def test_a_peer_that_dies_does_not_take_the_survivor_with_it(gpu_ctx, tmp_path):
    """Two ranks over RCCL (both on GPU 0, ranks as hosts), one good step each, then rank 1's process exits without a word.  The
    survivor's next step must come back with an error within the watchdog's deadline -- FFQ_E_TIMEOUT naming the stage, or RCCL's
    own asynchronous error for the dead peer --, ffq_shard_abort must return, and the process must end by itself: what a real
    node failure looks like to the ranks that are left (tests/rccl_peer_dies_worker.py; no torch.distributed.run around them --
    its agent would kill the survivor)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    idfile = str(tmp_path / "rccl.id")
    w = os.path.join(ROOT, "tests", "rccl_peer_dies_worker.py")
    p1 = subprocess.Popen([sys.executable, w, "1", idfile], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    p0 = subprocess.Popen([sys.executable, w, "0", idfile], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out0, _ = p0.communicate(timeout=300)
        out1, _ = p1.communicate(timeout=60)
    except subprocess.TimeoutExpired:
        p0.kill(); p1.kill()
        raise AssertionError("the survivor did not come back: " + (p0.stdout.read() if p0.stdout else "")[-2000:])
    assert "rank 1: first step ok" in out1, out1[-2000:]
    assert p0.returncode == 0 and "survivor ok" in out0, out0[-4000:]
    assert "rank 0: FFQTimeout" in out0 or "rank 0: FFQError" in out0, out0[-2000:]
