"""A rendezvous port for a launcher that is started a moment later: one BELOW the kernel's ephemeral range (32768-60999 here),
so that no outgoing connection or bind-to-0 of another test (RCCL's bootstrap and socket transport open dozens) can take it between
the probe and the launcher's listen -- `bind(("", 0))` hands out exactly such ports, and a full gpu suite lost one run to
EADDRINUSE that way (profiles/r06_b: test_bench_multi_rank_glue_on_rccl_with_one_rank)."""
import os
import random
import socket


def free_port():
    rng = random.Random(os.getpid() * 7919 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rng.randrange(15000, 30000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    raise RuntimeError("no free port between 15000 and 30000")
