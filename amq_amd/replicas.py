"""Multi-GPU = independent replicas: one process per GPU, one decode
stream per process, no collective inside the token loop.  The only communication
is a start/stop barrier and the max-over-ranks of the elapsed time (RCCL on
GPUs -- backend "nccl" is RCCL on ROCm -- gloo on CPU for tests)."""
import os
import subprocess
import tempfile
import time

import torch


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sys_root="/sys"):
    """NUMA node of every amdgpu card in PCI-address order (the order HIP enumerates devices in when nothing reorders them), -1 where the
    kernel does not say.  Read from /sys/class/drm/card*/device/numa_node -- no GPU runtime call (a launcher must not touch the GPU)."""
    cards = []
    drm = os.path.join(sys_root, "class", "drm")
    try:
        names = sorted(os.listdir(drm))
    except OSError:
        return []
    for name in names:
        if not name.startswith("card") or "-" in name:
            continue
        dev = os.path.join(drm, name, "device")
        try:
            if not os.path.basename(os.path.realpath(os.path.join(dev, "driver"))).startswith("amdgpu"):
                continue
            node = int(open(os.path.join(dev, "numa_node")).read().strip())
        except (OSError, ValueError):
            continue
        cards.append((os.path.basename(os.path.realpath(dev)), node))
    return [node for _, node in sorted(cards)]


def rank_cpu_set(local_rank, n_ranks, allowed=None, sys_root="/sys"):
    """The host cores rank ``local_rank`` of ``n_ranks`` should run on: the cores of ITS GPU's NUMA node (of those this process may use), divided
    evenly among the ranks whose GPUs share that node; without topology information (no amdgpu cards in sysfs, numa_node -1) an even share of
    the allowed cores.  Eight un-pinned host processes on a two-socket node otherwise migrate across sockets and -- rank 0's CPU-baseline leg
    aside, which runs at N = 1 only -- share cores.  Returns a non-empty set; never more than ``allowed``."""
    allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
    order = sorted(allowed)
    nodes = gpu_numa_nodes(sys_root)
    mine = nodes[local_rank] if local_rank < len(nodes) else -1
    if mine >= 0:
        try:
            node_cpus = _parse_cpulist(open(os.path.join(sys_root, "devices", "system", "node", f"node{mine}", "cpulist")).read()) & allowed
        except OSError:
            node_cpus = set()
        peers = [r for r in range(n_ranks) if r < len(nodes) and nodes[r] == mine]
        if node_cpus and local_rank in peers and len(node_cpus) >= len(peers):
            order, n_ranks, local_rank = sorted(node_cpus), len(peers), peers.index(local_rank)
    if len(order) < n_ranks:                       # fewer cores than ranks: everyone keeps what is allowed
        return set(order)
    per = len(order) // n_ranks
    return set(order[local_rank * per:(local_rank + 1) * per])


def pin_rank_cpus(local_rank=None, n_ranks=None):
    """Bind the calling rank to :func:`rank_cpu_set` (ranks started by another launcher, e.g. torch.distributed.run, call this themselves before
    they touch the GPU; :func:`launch_local` does it for its children).  No-op for one rank.  Returns the set applied (or None)."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    n_ranks = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if n_ranks is None else n_ranks
    if n_ranks <= 1:
        return None
    cpus = rank_cpu_set(local_rank, n_ranks)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return cpus


def launch_local(n, argv, env=None, timeout=None, sys_root="/sys"):
    """Start ``n`` ranks of ``argv`` (a full command line) as fresh child processes of THIS process, one per GPU:
    RANK / LOCAL_RANK / WORLD_SIZE (and the rendezvous, below) are set per child, stdout / stderr are inherited (rank 0
    prints the result line).  The caller must not have touched the GPU: children are started with Popen, never by
    re-executing a process that holds a device context.  Returns the largest exit code; if one rank fails the others
    are terminated by PID.
    Rendezvous: a FILE store in a fresh private directory (AMQ_RENDEZVOUS_FILE, read by :class:`Replicas`) -- no TCP port is
    picked here, so there is no window in which another process can take it between the probe and rank 0's bind.
    CPU placement: rank r runs on :func:`rank_cpu_set` (r, n) -- disjoint core sets, each inside the NUMA node of GPU r where sysfs says which
    (``sys_root``: where to read the topology from; tests hand in a made-up tree)."""
    rdv_dir = tempfile.mkdtemp(prefix="amq_rdv_")
    rdv = os.path.join(rdv_dir, "store")
    procs = []
    allowed = set(os.sched_getaffinity(0))
    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "AMQ_RENDEZVOUS_FILE": rdv})
        e.setdefault("MASTER_ADDR", "127.0.0.1")
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # each rank is bound to the cores of its GPU's NUMA node (its even share of them) BEFORE its interpreter starts: threads torch / the HIP
        # runtime create later inherit the mask
        cpus = rank_cpu_set(r, n, allowed, sys_root) if n > 1 else None
        procs.append(subprocess.Popen(list(argv), env=e, preexec_fn=(lambda c=cpus: os.sched_setaffinity(0, c)) if cpus else None))
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
