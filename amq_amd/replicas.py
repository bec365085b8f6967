"""Multi-GPU = independent replicas: one process per GPU, one decode
stream per process, no collective inside the token loop.  The only communication
is a start/stop barrier and the max-over-ranks of the elapsed time (RCCL on
GPUs -- backend "nccl" is RCCL on ROCm -- gloo on CPU for tests)."""
import os
import subprocess
import tempfile
import time

import torch


def launch_local(n, argv, env=None, timeout=None):
    """Start ``n`` ranks of ``argv`` (a full command line) as fresh child processes of THIS process, one per GPU:
    RANK / LOCAL_RANK / WORLD_SIZE (and the rendezvous, below) are set per child, stdout / stderr are inherited (rank 0
    prints the result line).  The caller must not have touched the GPU: children are started with Popen, never by
    re-executing a process that holds a device context.  Returns the largest exit code; if one rank fails the others
    are terminated by PID.
    Rendezvous: a FILE store in a fresh private directory (AMQ_RENDEZVOUS_FILE, read by :class:`Replicas`) -- no TCP port is
    picked here, so there is no window in which another process can take it between the probe and rank 0's bind."""
    rdv_dir = tempfile.mkdtemp(prefix="amq_rdv_")
    rdv = os.path.join(rdv_dir, "store")
    procs = []
    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "AMQ_RENDEZVOUS_FILE": rdv})
        e.setdefault("MASTER_ADDR", "127.0.0.1")
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(list(argv), env=e))
    t_end = None if timeout is None else time.monotonic() + timeout
    rc, live = 0, list(procs)
    try:
        rc = _reap(live, t_end)
    finally:
        for f in (rdv, rdv_dir):
            try:
                (os.remove if f == rdv else os.rmdir)(f)
            except OSError:
                pass
    return rc


def _reap(live, t_end):
    rc = 0
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            rc = max(rc, abs(code))
            if code != 0:                          # one rank died: the others would wait in the barrier forever
                for q in live:
                    q.terminate()
        if live and t_end is not None and time.monotonic() > t_end:
            for q in live:
                q.kill()
            rc = max(rc, 124)
            t_end = None                           # killed once; keep polling until they are reaped
        time.sleep(0.05)
    return rc


class Replicas:
    def __init__(self, backend=None, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend is None:
                backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
            kw = {"device_id": device} if backend == "nccl" else {}
            rdv = os.environ.get("AMQ_RENDEZVOUS_FILE")            # set by launch_local: file store, no TCP port involved
            if rdv:
                kw["init_method"] = "file://" + rdv
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def live_world_size(self):
        """ranks of the LIVE process group (1 without one) -- what the collectives below actually span"""
        return 1 if self.dist is None else int(self.dist.get_world_size())

    def live_backend(self):
        """backend name of the live process group ("nccl" is RCCL on ROCm), or "none" """
        return "none" if self.dist is None else str(self.dist.get_backend())

    def barrier(self):
        if self.dist is not None:
            if self.device is not None and self.device.type == "cuda":
                self.dist.barrier(device_ids=[self.local_rank])
            else:
                self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, value):
        """every rank's ``value`` (float), in rank order, on every rank"""
        if self.dist is None:
            return [float(value)]
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def timed(self, fn, steps, sync=lambda: None):
        """barrier + sync, run fn() `steps` times, sync + barrier; returns max-over-ranks seconds"""
        sync()
        self.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        self.last_local = time.perf_counter() - t0      # this rank's own time (before waiting for the others)
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
