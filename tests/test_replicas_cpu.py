"""CPU: the N>1 path of bench.py (replicas: barrier + max-over-ranks timing) with gloo, world_size 2."""
import os
import socket
import time

import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update({"WORLD_SIZE": str(world), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from amq_amd.replicas import Replicas
    r = Replicas(backend="gloo")
    assert r.world == world and r.rank == rank
    calls = []
    t = r.timed(lambda: (calls.append(1), time.sleep(0.02 * (rank + 1))), steps=5)
    whole_job = world * 5 / t            # tokens/s analogue: every rank did 5 steps
    q.put((rank, len(calls), t, whole_job, r.max_over_ranks(rank)))
    r.close()


def test_two_replicas_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (r0, n0, t0, v0, m0), (r1, n1, t1, v1, m1) = res
    assert n0 == n1 == 5
    assert abs(t0 - t1) < 1e-9 and t0 >= 5 * 0.04          # both report the slower rank's time
    assert m0 == m1 == 1.0
    assert abs(v0 - 2 * 5 / t0) < 1e-9


def test_single_process_needs_no_process_group():
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    from amq_amd.replicas import Replicas
    r = Replicas()
    assert r.world == 1 and r.dist is None
    assert r.max_over_ranks(3.5) == 3.5
    assert r.timed(lambda: None, 3) >= 0.0
