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


_RANK_SCRIPT = """
import json, os, sys, time
sys.path.insert(0, {root!r})
from amq_amd.replicas import Replicas
r = Replicas(backend="gloo")
t = r.timed(lambda: time.sleep(0.01 * (r.rank + 1)), steps=4)
per = r.gather(4 / r.last_local)
if r.rank == 0:
    print(json.dumps({{"n": r.world, "t": t, "per_rank": per, "value": r.world * 4 / t}}), flush=True)
r.close()
"""


def test_launch_local_spawns_ranks(tmp_path):
    """what `bench.py --gpus N` does when no launcher set WORLD_SIZE: N fresh children, one result line from rank 0"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=root))
    drv = tmp_path / "drv.py"
    drv.write_text(f"import sys; sys.path.insert(0, {root!r})\nfrom amq_amd.replicas import launch_local\n"
                   f"sys.exit(launch_local(2, [sys.executable, {str(script)!r}], timeout=120))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=180, env=env)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n"] == 2 and len(d["per_rank"]) == 2 and d["per_rank"][0] > d["per_rank"][1]
    assert d["t"] >= 4 * 0.02 and abs(d["value"] - 2 * 4 / d["t"]) < 1e-6
    # a failing rank fails the launch (and does not hang it)
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nsys.exit(3) if os.environ['RANK'] == '1' else time.sleep(30)\n")
    drv.write_text(f"import sys; sys.path.insert(0, {root!r})\nfrom amq_amd.replicas import launch_local\n"
                   f"sys.exit(launch_local(2, [sys.executable, {str(bad)!r}], timeout=60))\n")
    out = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=90, env=env)
    assert out.returncode != 0


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` with no launcher starts 2 ranks itself (here, without a GPU, both refuse loudly and the
    launch fails instead of silently benchmarking one device)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert out.stderr.count("needs a GPU") >= 1          # (the second rank may be terminated before it prints)
    # BASELINE.json configs[4] (one 70B decode stream per rank) takes the same launch path
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "5", "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert out.stderr.count("needs a GPU") >= 1
    # a launcher that started a different number of ranks is an error, not a silent 1-GPU run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env2)
    assert out.returncode != 0 and "launcher started 1 rank" in out.stderr


def test_single_process_needs_no_process_group():
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    from amq_amd.replicas import Replicas
    r = Replicas()
    assert r.world == 1 and r.dist is None
    assert r.max_over_ranks(3.5) == 3.5
    assert r.timed(lambda: None, 3) >= 0.0


def test_synthetic_tokenizer_round_trip():
    """the TTFT stand-in tokenizer: ids -> text -> ids is the identity, truncation cuts, decode of one id gives one word
    (the three calls of the reference's TTFT loop, amq/utils/speed.py:193, 214-217)"""
    import torch
    from amq_amd.speed import SyntheticTokenizer
    tok = SyntheticTokenizer(32000)
    ids = torch.randint(0, 31999, (64,), generator=torch.Generator().manual_seed(0))
    text = tok.decode(ids)
    back = tok(text, return_tensors="pt", truncation=True, max_length=64).input_ids
    assert back.shape == (1, 64) and torch.equal(back[0], ids)
    assert tok(text, max_length=10).input_ids.shape == (1, 10)
    assert tok.decode([123]) == "t123"


def _fake_sysfs(root, gpu_nodes, node_cpus):
    """a /sys tree with amdgpu cards (PCI order = list order) on the given NUMA nodes and nodes with the given cpulists"""
    for i, node in enumerate(gpu_nodes):
        pci = root / "devices" / f"pci0000:{i:02x}" / f"0000:{i:02x}:00.0"
        pci.mkdir(parents=True)
        (pci / "numa_node").write_text(f"{node}\n")
        drv = root / "bus" / "pci" / "drivers" / "amdgpu"
        drv.mkdir(parents=True, exist_ok=True)
        (pci / "driver").symlink_to(drv)
        card = root / "class" / "drm" / f"card{7 - i}"          # card numbers need not follow PCI order
        card.mkdir(parents=True)
        (card / "device").symlink_to(pci)
    for node, cpus in node_cpus.items():
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")


def test_rank_cpu_sets_follow_the_gpus_numa_nodes(tmp_path):
    """8 GPUs on two sockets: rank r gets an even share of the cores of GPU r's node; shares are disjoint, inside the allowed mask, and
    fall back to an even split of the allowed cores when sysfs has no topology"""
    from amq_amd.replicas import gpu_numa_nodes, rank_cpu_set
    _fake_sysfs(tmp_path, [0, 0, 0, 0, 1, 1, 1, 1], {0: "0-63,128-191", 1: "64-127,192-255"})
    assert gpu_numa_nodes(str(tmp_path)) == [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = set(range(256))
    sets = [rank_cpu_set(r, 8, allowed, str(tmp_path)) for r in range(8)]
    node0 = set(range(0, 64)) | set(range(128, 192))
    for r, s in enumerate(sets):
        assert len(s) == 32 and s < allowed
        assert (s <= node0) == (r < 4)                       # ranks 0-3 on node 0's cores, 4-7 on node 1's
    assert len(set().union(*sets)) == 256                    # disjoint, nothing left over
    # a restricted mask (a container's cpuset) is respected
    few = set(range(0, 16)) | set(range(64, 80))
    sets = [rank_cpu_set(r, 8, few, str(tmp_path)) for r in range(8)]
    assert all(len(s) == 4 and s < few for s in sets) and len(set().union(*sets)) == 32
    # no topology: even split of what is allowed
    empty = tmp_path / "none"
    empty.mkdir()
    sets = [rank_cpu_set(r, 4, set(range(8)), str(empty)) for r in range(4)]
    assert sets == [{0, 1}, {2, 3}, {4, 5}, {6, 7}]
    assert rank_cpu_set(0, 4, {5}, str(empty)) == {5}        # fewer cores than ranks: everyone keeps the mask


def test_launch_local_pins_each_rank(tmp_path):
    """launch_local binds every child before its interpreter starts: masks are disjoint strict subsets of the launcher's"""
    import json, subprocess, sys
    host = sorted(os.sched_getaffinity(0))
    if len(host) < 2:
        import pytest
        pytest.skip("needs two cores")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "aff.py"
    script.write_text("import json, os\nprint(json.dumps({'rank': int(os.environ['RANK']), 'cpus': sorted(os.sched_getaffinity(0))}), flush=True)\n")
    drv = tmp_path / "drv.py"
    drv.write_text(f"import sys; sys.path.insert(0, {root!r})\nfrom amq_amd.replicas import launch_local\n"
                   f"sys.exit(launch_local(2, [sys.executable, {str(script)!r}], timeout=60))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr
    got = {d["rank"]: set(d["cpus"]) for d in map(json.loads, [l for l in out.stdout.splitlines() if l.startswith("{")])}
    assert set(got) == {0, 1}
    assert got[0] < set(host) and got[1] < set(host) and not (got[0] & got[1])


def test_bench_extras_run_after_the_timed_region():
    """rank 0's CPU-baseline / parity / extras legs start only after `rep.timed` has closed its barrier on every rank (they would share cores and
    HBM with the other ranks' timed steps otherwise): in bench.run_decode nothing but the timed call sits between warm-up and `rep.gather`"""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "run_decode")
    src = ast.get_source_segment(open(os.path.join(root, "bench.py")).read(), fn)
    timed = src.index("rep.timed(")
    for leg in ("gemv_layer_table(", "beyond_the_metric(", "mfma_roofline(", "parity_gate(", "cpu_baseline()", "gemv_roofline("):
        assert src.index(leg) > timed, leg
    # ... and only rank 0 at N = 1 runs them
    assert "if rank != 0:\n        return" in src and src.count("n_gpus == 1") >= 3


_RANK8_SCRIPT = """
import json, os, sys, time
sys.path.insert(0, {root!r})
from amq_amd.replicas import Replicas
r = Replicas(backend="gloo")
t = r.timed(lambda: time.sleep(0.005 * (1 + r.rank % 3)), steps=3)
left_timed = time.time()
per = r.gather(3 / r.last_local)
ends = r.gather(left_timed)
if r.rank == 0:
    extras_start = time.time()                      # (bench.py: rank 0's CPU-baseline / parity / extras legs start here)
    time.sleep(0.05)
r.barrier()                                         # the closing barrier: nobody tears the group down under rank 0's extras
closed = time.time()
print(json.dumps({{"rank": r.rank, "world": r.live_world_size(), "cpus": sorted(os.sched_getaffinity(0)), "per_rank": per, "t": t,
                  "ends": ends, "extras_start": extras_start if r.rank == 0 else None, "closed": closed}}), flush=True)
r.close()
"""


def test_eight_ranks_over_a_two_socket_node(tmp_path):
    """BASELINE.json configs[4] is 8 decode streams on 8 GPUs of one node; no such node was offered to any round, so the launch path is
    rehearsed here at its real width: launch_local starts 8 ranks over a made-up sysfs of 8 amdgpu cards on two sockets -- every rank bound inside
    its GPU's NUMA node, masks disjoint -- they meet through the file-store rendezvous (gloo standing in for RCCL), the timed region reports the
    slowest rank's time on every rank, the gather returns 8 rates in rank order, and rank 0's extra legs start only after every rank has left
    the timed region and finish before anybody passes the closing barrier."""
    import json, subprocess, sys
    host = sorted(os.sched_getaffinity(0))
    if len(host) < 8:
        import pytest
        pytest.skip("needs eight cores")
    host = host[:8]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "sys"
    lst = lambda cs: ",".join(str(c) for c in cs)
    _fake_sysfs(fake, [0, 0, 0, 0, 1, 1, 1, 1], {0: lst(host[:4]), 1: lst(host[4:])})
    script = tmp_path / "rank8.py"
    script.write_text(_RANK8_SCRIPT.format(root=root))
    drv = tmp_path / "drv.py"
    drv.write_text(f"import os, sys; sys.path.insert(0, {root!r})\nos.sched_setaffinity(0, {set(host)!r})\nfrom amq_amd.replicas import launch_local\n"
                   f"sys.exit(launch_local(8, [sys.executable, {str(script)!r}], timeout=240, sys_root={str(fake)!r}))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = sorted((json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in rows] == list(range(8)) and all(d["world"] == 8 for d in rows)
    masks = [set(d["cpus"]) for d in rows]
    assert all(len(m) == 1 for m in masks) and len(set().union(*masks)) == 8           # one core each, disjoint
    assert all((m <= set(host[:4])) == (r < 4) for r, m in enumerate(masks))           # ranks 0-3 on socket 0's cores, 4-7 on socket 1's
    assert all(len(d["per_rank"]) == 8 and d["per_rank"] == rows[0]["per_rank"] for d in rows)
    assert all(abs(d["t"] - rows[0]["t"]) < 1e-9 for d in rows) and rows[0]["t"] >= 3 * 0.015   # the slowest rank's time, on every rank
    pr = rows[0]["per_rank"]
    assert pr[0] > pr[2] and pr[3] > pr[5]                                             # rank order kept (ranks 2, 5 sleep 3 x longer)
    assert rows[0]["extras_start"] >= max(rows[0]["ends"]) - 1e-3                      # extras begin after EVERY rank left the timed region
    assert all(d["closed"] >= rows[0]["extras_start"] + 0.05 - 1e-3 for d in rows)     # ... and end before anyone passes the closing barrier


def test_bench_rehearses_the_n_rank_launch_path():
    """`bench.py --gpus N --config 5 --rehearse-ranks`: the script's own launch path at N ranks (launch_local children, placement, rendezvous,
    timed region, gather, rank 0's line) with gloo standing in for RCCL and a sleep for the decode step -- the line says it is a rehearsal and
    carries no value; what is checked is what an 8-GPU node will need: one line, N per-rank rates, a live process group of N ranks"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AMQ_RENDEZVOUS_FILE")}
    for n in (2, 8):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--config", "5", "--steps", "4", "--warmup", "1",
                              "--rehearse-ranks"], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["rehearsal"] is True and d["value"] is None and d["n_gpus"] == n
        assert d["rccl_world_size"] == n and len(d["per_rank_tokens_per_s"]) == n and len(d["per_rank_host_cores"]) == n
        assert d["config"]["parallelism"] == f"replicas x{n}" and "configs[4]" in d["config"]["workload"]
    # under an outer launcher (torch.distributed.run sets WORLD_SIZE / RANK / MASTER_*): the existing ranks are used, a mismatch is an error
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29591", os.path.join(root, "bench.py"), "--gpus", "2", "--config", "5", "--steps", "3", "--warmup", "0",
                          "--rehearse-ranks"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["rccl_world_size"] == 2 and len(d["per_rank_tokens_per_s"]) == 2
