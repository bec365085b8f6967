#!/usr/bin/env python3
"""ISA-level check of every hand-counted wait in the product kernels (CPU only).

The kernels built on LDS-DMA (`global_load_lds` / `buffer_load ... lds` from inline asm) wait for their transfers with `s_waitcnt vmcnt(N)`
where N is COUNTED in the source: "this phase issued 3 pieces, the one before 5, so with 8 in flight the tile before has landed".  The compiler
sees neither the transfers nor the count, and a load it merges, splits or moves (round 5: three dword loads merged into one; the rows staged by
LDS-DMA then waited one tile short at 3 bit) silently breaks the arithmetic.  Every such wait goes through `AMQ_WAIT_VM(name, n, spec)`
(amq_amd/csrc/amq_common.cuh), which leaves `; AMQ_WAIT id=<name> n=<n> from=<site>:<ops> ...` as a comment in the device assembly the Makefile
keeps under amq_amd/csrc/asm/ (the text the objects were assembled from).  This tool rebuilds each kernel's control-flow graph from that text and
checks, for every wait W and every `from=F:c`:

    on EVERY path from a site named F (`entry`: the kernel's first instruction; otherwise an AMQ_WAIT / AMQ_MARK with that id) to W that does
    not pass another site named in W's from-list (or another W), the number of vector-memory instructions -- everything that counts on vmcnt:
    global / buffer / flat / scratch loads, stores, atomics, LDS-DMA -- is AT LEAST c.

vmcnt retires in issue order, so `s_waitcnt vmcnt(n)` covers a transfer exactly when at least n younger operations were issued behind it: FEWER
operations than the source counts on some path is the unsafe direction and a VIOLATION.  More than c (reported as `min..max` in the table, and as a
note) only makes the wait stricter than planned -- a performance matter; most such paths are static only: the structurizer's flag blocks leave
edges in the graph (an epilogue "falling into" the other wave role's code) that no wave takes.  The tests pin the MINIMUM to the source's count
for every product site (tests/test_waits_cpu.py), so a count that grows on every path is seen too.  Every wait with a from-list must be reached
from at least one of its sites, and an inline-asm `s_waitcnt` without a tag is refused.

usage: check_waits.py [--asm DIR] [--json OUT] [-q]      exit status 1 on any violation
"""
import argparse
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM_DIR = os.path.join(ROOT, "amq_amd", "csrc", "asm")
VMEM_PREFIX = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "buffer_wbl2", "buffer_inv",
               "flat_load", "flat_store", "flat_atomic", "scratch_load", "scratch_store", "tbuffer_load", "tbuffer_store", "image_")
INF = float("inf")


class Fn:
    """one function of a device assembly file: instructions, successor lists, tagged sites"""

    def __init__(self, name):
        self.name = name
        self.ins = []           # (mnemonic, operand text, line number)
        self.succ = []
        self.vmem = []
        self.sites = {}         # instruction index -> dict(kind, id, n, froms)
        self.untagged = []      # inline-asm s_waitcnt without a tag: line numbers
        self.labels = {}
        self._pending = []      # (instruction index, label) branches to resolve


def parse(path):
    fns, cur, in_asm = [], None, False
    for ln, raw in enumerate(open(path, errors="replace"), 1):
        line = raw.rstrip("\n")
        s = line.strip()
        if not s:
            continue
        m = re.match(r"^([A-Za-z_$][\w$.]*):", line)
        if m and not m.group(1).startswith(".L"):
            cur = Fn(m.group(1))
            fns.append(cur)
            continue
        if cur is None:
            continue
        if re.match(r"^\.Lfunc_end\d+:", s):
            cur = None
            continue
        m = re.match(r"^(\.L[\w$.]+):", s)
        if m:
            cur.labels[m.group(1)] = len(cur.ins)
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s.startswith(";") or s.startswith("."):
            continue
        code, _, comment = s.partition(";")
        parts = code.split(None, 1)
        if not parts:
            continue
        mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        i = len(cur.ins)
        cur.ins.append((mn, ops, ln))
        cur.vmem.append(mn.startswith(VMEM_PREFIX))
        tag = re.search(r"AMQ_(WAIT|MARK)\s+id=(\S+)(.*)", comment)
        if tag:
            rest = tag.group(3)
            n = re.search(r"\bn=(\d+)", rest)
            froms = [(f, op or "==", int(c)) for f, op, c in re.findall(r"from=([\w.]+):(>=)?(\d+)", rest)]
            cur.sites[i] = dict(kind=tag.group(1), id=tag.group(2), n=int(n.group(1)) if n else None, froms=froms, line=ln)
        elif in_asm and mn == "s_waitcnt":
            cur.untagged.append(ln)
        if mn == "s_branch":
            cur._pending.append((i, ops.strip(), False))
        elif mn.startswith("s_cbranch"):
            cur._pending.append((i, ops.split(",")[-1].strip(), True))
    for f in fns:
        n = len(f.ins)
        f.succ = [[i + 1] if i + 1 < n else [] for i in range(n)]
        for i, (mn, _, _) in enumerate(f.ins):
            if mn in ("s_endpgm", "s_setpc_b64", "s_trap") or mn.startswith("s_endpgm"):
                f.succ[i] = []
        for i, lab, cond in f._pending:
            tgt = f.labels.get(lab)
            if tgt is None:
                raise SystemExit(f"{path}: {f.name}: branch to unknown label {lab}")
            f.succ[i] = ([i + 1] if cond and i + 1 < n else []) + ([tgt] if tgt < n else [])
    return fns


def path_counts(f, src, dst, stop):
    """(min, max) vector-memory instructions strictly between instruction src and instruction dst over all paths that pass no instruction of
    `stop` on the way; None if dst is not reached.  max = inf if a cycle on the way issues vector-memory instructions."""
    fwd, stack = {src}, [src]
    while stack:
        u = stack.pop()
        for v in f.succ[u]:
            if v in fwd or (v in stop and v != dst):
                continue
            fwd.add(v)
            if v != dst:
                stack.append(v)
    if dst not in fwd:
        return None
    pred = defaultdict(list)
    for u in fwd:
        if u == dst:
            continue
        for v in f.succ[u]:
            if v in fwd and v != src:             # (a path that comes back to its own starting site starts over there)
                pred[v].append(u)
    back, stack = {dst}, [dst]
    while stack:
        v = stack.pop()
        for u in pred[v]:
            if u not in back:
                back.add(u)
                stack.append(u)
    nodes = back                                  # on some src -> dst path
    w = lambda v: 1 if (f.vmem[v] and v != src and v != dst) else 0
    succ = {u: [v for v in f.succ[u] if v in nodes and v != src] if u != dst else [] for u in nodes}
    # strongly connected components (iterative Tarjan)
    index, low, onst, comp, st, order = {}, {}, set(), {}, [], []
    for root in nodes:
        if root in index:
            continue
        work = [(root, 0)]
        while work:
            u, k = work.pop()
            if k == 0:
                index[u] = low[u] = len(index)
                st.append(u)
                onst.add(u)
            recursed = False
            for j in range(k, len(succ[u])):
                v = succ[u][j]
                if v not in index:
                    work.append((u, j + 1))
                    work.append((v, 0))
                    recursed = True
                    break
                if v in onst:
                    low[u] = min(low[u], index[v])
            if recursed:
                continue
            if low[u] == index[u]:
                c = len(order)
                members = []
                while True:
                    v = st.pop()
                    onst.discard(v)
                    comp[v] = c
                    members.append(v)
                    if v == u:
                        break
                order.append(members)
            if work:
                p = work[-1][0]
                low[p] = min(low[p], low[u])
    cyc_vmem = False
    cw = []
    for members in order:
        cyclic = len(members) > 1 or members[0] in succ[members[0]]
        tot = sum(w(v) for v in members)
        if cyclic and tot:
            cyc_vmem = True
        cw.append(tot)
    # Tarjan emits components in reverse topological order: successors first
    lo, hi = {}, {}
    for c, members in enumerate(order):
        outs = {comp[v] for u in members for v in succ[u]} - {c}
        if comp[dst] == c:
            lo[c] = hi[c] = cw[c]
        else:
            lo[c] = cw[c] + min(lo[o] for o in outs)
            hi[c] = cw[c] + max(hi[o] for o in outs)
    c0 = comp[src]
    if cyc_vmem:
        # the component sums count a cycle's instructions once each: take the true minimum from a 0-1 shortest path instead
        from collections import deque
        dist, dq = {src: 0}, deque([src])
        while dq:
            u = dq.popleft()
            for v in succ[u]:
                d = dist[u] + w(v)
                if d < dist.get(v, INF):
                    dist[v] = d
                    (dq.appendleft if w(v) == 0 else dq.append)(v)
        return dist[dst], INF
    return lo[c0], hi[c0]


def check_fn(f):
    """-> (rows, errors): one row per (wait, from) pair"""
    rows, errs = [], []
    by_id = defaultdict(list)
    for i, s in f.sites.items():
        by_id[s["id"]].append(i)
    for i, s in sorted(f.sites.items()):
        if s["kind"] != "WAIT" or not s["froms"]:
            continue
        names = {fr for fr, _, _ in s["froms"]} | {s["id"]}
        stop = {j for nm in names for j in by_id.get(nm, [])}
        covered = False
        for fr, op, c in s["froms"]:
            srcs = [0] if fr == "entry" else by_id.get(fr, [])
            for src in srcs:
                got = path_counts(f, src, i, stop - {src})
                if got is None:
                    continue
                covered = True
                lo, hi = got
                ok = lo >= c
                rows.append(dict(kernel=f.name, wait=s["id"], n=s["n"], line=s["line"], frm=fr, op=op, want=c, min=lo,
                                 max=None if hi == INF else hi, ok=ok))
                if not ok:
                    errs.append(f"{f.name}: wait {s['id']} (line {s['line']}, vmcnt({s['n']})): {lo}..{'inf' if hi == INF else hi} vector-memory "
                                f"instructions since {fr}, the source counts {c}")
        if not covered:
            errs.append(f"{f.name}: wait {s['id']} (line {s['line']}): none of its from-sites {[x[0] for x in s['froms']]} reaches it")
    for ln in f.untagged:
        errs.append(f"{f.name}: inline-asm s_waitcnt without an AMQ_WAIT tag at line {ln}")
    return rows, errs


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return [re.sub(r"\(.*", "", o.replace("void amq::", "")) for o in out]
    except Exception:
        return list(names)


def run(asm_dir=ASM_DIR, files=None):
    if files is None:
        files = sorted(os.path.join(asm_dir, f) for f in os.listdir(asm_dir) if f.endswith(".s")) if os.path.isdir(asm_dir) else []
    rows, errs, n_waits, n_kernels = [], [], 0, 0
    for p in files:
        for f in parse(p):
            if not f.sites and not f.untagged:
                continue
            n_kernels += 1
            n_waits += sum(1 for s in f.sites.values() if s["kind"] == "WAIT")
            r, e = check_fn(f)
            for x in r:
                x["file"] = os.path.basename(p)
            rows += r
            errs += [os.path.basename(p) + ": " + x for x in e]
    return dict(files=[os.path.basename(p) for p in files], kernels=n_kernels, waits=n_waits, rows=rows, errors=errs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=ASM_DIR)
    ap.add_argument("--json")
    ap.add_argument("-q", action="store_true")
    a = ap.parse_args()
    res = run(a.asm)
    if not res["files"]:
        raise SystemExit(f"no assembly under {a.asm}: build first (make -C amq_amd/csrc)")
    if not a.q:
        agg = defaultdict(lambda: [0, INF, 0, 0])
        for r in res["rows"]:
            k = (r["file"], r["wait"], r["n"], r["frm"], r["op"], r["want"])
            g = agg[k]
            g[0] += 1
            g[1] = min(g[1], r["min"])
            g[2] = max(g[2], r["max"] if r["max"] is not None else 10 ** 9)
            g[3] += 0 if r["ok"] else 1
        print(f"{'file':22s} {'wait':20s} {'vmcnt':>5s} {'since':14s} {'source':>7s} {'ISA min..max':>13s} {'sites':>6s} {'bad':>4s}")
        for (fl, wt, n, fr, op, want), (cnt, lo, hi, bad) in sorted(agg.items()):
            print(f"{fl:22s} {wt:20s} {n!s:>5s} {fr:14s} {('>=' if op == '>=' else '') + str(want):>7s} {str(lo) + '..' + ('inf' if hi >= 10 ** 9 else str(hi)):>13s} {cnt:6d} {bad:4d}")
        print(f"{res['waits']} tagged waits in {res['kernels']} kernels of {len(res['files'])} files; {len(res['errors'])} violation(s)")
    for e in res["errors"][:50]:
        print("VIOLATION:", e, file=sys.stderr)
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(dict(kernels=res["kernels"], waits=res["waits"], errors=res["errors"],
                           rows=[{k: v for k, v in r.items()} for r in res["rows"]]), fh)
    sys.exit(1 if res["errors"] else 0)


if __name__ == "__main__":
    main()
