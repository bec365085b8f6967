import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench, os
from amq_amd.llama import QuantLlama
if os.environ.get('FRAG_ROWS'):
    QuantLlama.FRAG_ROWS = tuple(int(v) for v in os.environ['FRAG_ROWS'].split(','))
SS = tuple(int(v) for v in os.environ.get('PREFILL_S', '16,64,128,256,512,1024,2048').split(','))
m, a, usage = bench.build_model(torch.device('cuda:0'), seed=0, max_seq=2200)
for S in SS:
    ids = torch.randint(0, m.vocab - 1, (S,), generator=torch.Generator().manual_seed(0)).to(m.dev)
    for _ in range(2):
        m.prefill(ids)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m.prefill(ids); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"prefill S={S}: {ts[2]*1e3:.2f} ms  ({S/ts[2]:.0f} tokens/s)", flush=True)
