"""BASELINE configs[4] at full size on ONE GPU, and the RCCL transport under test.

configs[4]: 100 GiB synthetic 150 bp (107 374 182 146 B, 333 460 193 records) sharded into 8
byte ranges with the chunk-edge stitch.  An 8-GPU node is not available to these tests; what one
MI355X can show is (a) the whole protocol at full size -- 8 logical ranks as threads, each with its
12.5 GiB range resident, its own context, halos handed off by the in-process transport, the HIP
engine on every range, offsets past 2^36 -- and (b) the product's transport, the library's own RCCL
one (ffq_shard_create), initialised at world size 1: two communicators, the gather of the words,
pipelined lanes, a re-gather, send / recv to itself (with peers: tests/test_multigpu.py).
Semantics matched: /root/reference/src/fastqandfurious.py:251-279 (the record chain, here cut into
ranges: every rank's rows must be exactly the rows of the one-range scan that fall into its range).
"""
import os
import subprocess
import sys
import threading

import pytest
import torch

from conftest import ROOT

TOTAL_100G = 107374182146            # floor(100 GiB / 322) records of 322 B
RECORDS_100G = 333460193


@pytest.mark.gpu
@pytest.mark.parametrize("native", (True,))
def test_config5_eight_logical_ranges_full_size(gpu_ctx, native):
    """Every step behind the C ABI (ffq_shard_step_submit / _wait, in-process transport): the product's step with
    device copies in place of RCCL."""
    from fastqandfurious_amd import hip, sharded
    free, _tot = torch.cuda.mem_get_info()
    if free < 190 * (1 << 30):
        pytest.skip("needs ~170 GiB of free HBM (100 GiB resident + index + tables)")
    world = 8
    lw = sharded.LocalWorld(world)
    dev = torch.device("cuda", 0)
    errors, outs = [None] * world, [None] * world
    per = (100 << 30) // world

    def work(rank):
        try:
            ctx = hip.Context(0)
            sh = sharded.SyntheticShard(ctx, "single", per, rank, world, dev, transport=lw.transport(rank),
                                        total_records=RECORDS_100G, native=native)
            ctx.reserve(sh.ext.numel())
            table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
            out = sh.scan(table)                 # halo hand-off + scan + cut + the eight-word all_gather
            assert out.res.path == 3 and out.rounds == 0
            sh.verify(table, out)                # every row of the range against the generator's closed form
            # hand-off words: my first record / the first record past my right edge, as the chain sees them
            k0, k1 = -(-sh.own_lo // 322), -(-sh.own_hi // 322)
            assert out.first_pos == k0 * 322
            assert out.exit_pos == (k1 * 322 if rank < world - 1 else sharded.NONE_POS)
            assert out.record_base == k0 and out.n_own_records == k1 - k0
            outs[rank] = (out.total_records, out.record_base, out.n_own_records, sh.n_own_bytes, sh.own_lo,
                          int(table[out.row_hi - 1, 5].item()))
            ctx.close()
        except BaseException as e:   # noqa: BLE001
            errors[rank] = e
            lw.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)
            and "another logical rank failed" not in str(e)]
    if real:
        raise real[0]
    assert sum(o[3] for o in outs) == TOTAL_100G
    assert all(o[0] == RECORDS_100G for o in outs)
    assert sum(o[2] for o in outs) == RECORDS_100G
    assert [o[1] for o in outs] == [sum(x[2] for x in outs[:r]) for r in range(world)]      # global ordinals
    assert outs[-1][5] == TOTAL_100G - 1 and outs[-1][4] > (1 << 36)                           # the last pos5; offsets past 2^36
    torch.cuda.empty_cache()


from _ports import free_port as _free_port      # noqa: E402  (below the ephemeral range: tests/_ports.py)


def _run_worker(mode, timeout):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_worker.py"), mode], env=env, cwd=ROOT,
                              capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        return e


@pytest.mark.gpu
def test_native_step_on_rccl_world1(gpu_ctx):
    """The library's own step (ffq_shard_*) on its RCCL transport at world size 1: two communicators from one unique id,
    steps plain and pipelined over two lanes, the gather of the hand-off words, comm figures in the result; and
    ncclSend / ncclRecv to itself in one group on the hand-off stream (ffq_shard_self_exchange)."""
    r = _run_worker("native", 300)
    assert not isinstance(r, subprocess.TimeoutExpired), "nccl worker timed out"
    if r.returncode == 3:
        pytest.skip("RCCL refuses send/recv to self: " + r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "native step on rccl ok" in r.stdout
