"""Worker of tests/test_multigpu.py::test_a_peer_that_dies...: two processes on GPU 0 (ranks as hosts, NCCL_HOSTID), the library's
RCCL transport between them, NO torch.distributed.run around them (its agent would kill the survivor).  Both take one good step;
then rank 1 exits without a word; rank 0's next step must come back -- FFQTimeout from the watchdog, or the asynchronous error
RCCL reports for the dead peer -- within the deadline, abort() must return within ITS deadline (drained, or saying that ncclCommAbort itself is still busy), and the
process must be able to report and leave."""
import faulthandler, os, sys, time
rank = int(sys.argv[1]); idfile = sys.argv[2]
faulthandler.dump_traceback_later(60, repeat=False, file=sys.stderr, exit=False)       # (where a rank that does not come back is stuck)
os.environ["NCCL_HOSTID"] = "ffq-rank-as-host-%d" % rank
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import hip, sharded, synth
from test_sharded import bounds_for

world = 2
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if rank == 0:
    uid = hip.shard_unique_id()
    with open(idfile + ".tmp", "wb") as fh:
        fh.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    for _ in range(600):
        if os.path.exists(idfile):
            break
        time.sleep(0.1)
    uid = open(idfile, "rb").read()
stream = synth.wrapped(0, 12000, seed=43)[0]
bounds = bounds_for(stream.size, world, 0, 48)
lo, hi = bounds[rank], bounds[rank + 1]
ctx = hip.Context(0)
sc = sharded.NativeShardScanner(ctx, bounds, rank, world, unique_id=uid)
sc.sh.set_timeout(6.0)
tail, head = sc.halo()
ext = torch.zeros(tail + (hi - lo) + head + 64, dtype=torch.uint8, device=dev)
ext[tail:tail + hi - lo] = torch.from_numpy(stream[lo:hi].copy()).to(dev)
tab = torch.empty((stream.size // 40 + 64, 6), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
o = sc.scan(ext, tail, head, tab)
assert o.comm["nranks"] == 2 and o.comm["handoff_bytes"] > 0
print("rank %d: first step ok (%d own records of %d)" % (rank, o.n_own_records, o.total_records), flush=True)
if rank == 1:
    os._exit(0)                 # gone: no destroy, no abort, no goodbye
time.sleep(1.0)
ext[:tail].zero_(); ext[tail + hi - lo:].zero_(); torch.cuda.synchronize()
t0 = time.perf_counter()
try:
    sc.scan(ext, tail, head, tab)
    print("rank 0: the step after the peer's death CAME BACK", flush=True)
    sys.exit(5)
except hip.FFQError as e:
    waited = time.perf_counter() - t0
    print("rank 0: %s after %.1f s: %s" % (type(e).__name__, waited, e), flush=True)
    assert waited < 30, waited
t0 = time.perf_counter()
drained = sc.abort()
took = time.perf_counter() - t0
print("rank 0: abort -> %s in %.1f s (%s)" % (drained, took, "" if drained else hip.lib().ffq_last_error().decode()), flush=True)
assert took < 50, took
sc.close()                       # (a leaked shard: host objects only)
if drained:
    ctx.close()
print("survivor ok", flush=True)
sys.stdout.flush()
os._exit(0)                      # (RCCL's teardown of a communicator whose peer is gone may hold the device for minutes)
