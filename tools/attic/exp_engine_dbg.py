#!/usr/bin/env python3
"""engine debugging: where do the barrier words of a captured step get their contents from?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amq_amd import arch
from amq_amd.llama import QuantLlama

dev = torch.device("cuda:0")
cfg = dict(arch._cfg(2, 512, 1024, 4, 2, 1, vocab=1024))


def words(m, tag):
    torch.cuda.synchronize()
    s = m.engine.sync
    print(tag, "top/grp0/gen0/err", [hex(int(s[i]) & 0xFFFFFFFF) for i in (0, 64, 33 * 64, 65 * 64, 65 * 64 + 1)],
          "ptr sync %x image %x scratch %x" % (s.data_ptr(), m.engine.image.data_ptr(), m.engine.scratch.data_ptr()), flush=True)


m = QuantLlama(cfg, None, device="cuda:0", max_seq=16, seed=4, engine=True)
words(m, "after init      ")
m.prefill(torch.randint(0, 1024, (13,), generator=torch.Generator().manual_seed(1)).to(dev))
words(m, "after prefill   ")
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    m.engine.step()
    side.synchronize()
    words(m, "after eager/side")
    m.engine.step()
    side.synchronize()
    words(m, "2nd eager/side  ")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        m.engine.step()
torch.cuda.current_stream(dev).wait_stream(side)
words(m, "after capture   ")
g.replay()
words(m, "after replay 1  ")
g.replay()
words(m, "after replay 2  ")
m.engine.step()
words(m, "eager default   ")
g.replay()
words(m, "after replay 3  ")
