"""Multi-GPU = independent replicas: one process per GPU, one decode
stream per process, no collective inside the token loop.  The only communication
is a start/stop barrier and the max-over-ranks of the elapsed time (RCCL on
GPUs -- backend "nccl" is RCCL on ROCm -- gloo on CPU for tests)."""
import os
import time

import torch


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
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

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

    def timed(self, fn, steps, sync=lambda: None):
        """barrier + sync, run fn() `steps` times, sync + barrier; returns max-over-ranks seconds"""
        sync()
        self.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
