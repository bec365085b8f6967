"""CPU-only: the safety net under the hand-counted waits (SURVEY.md 5 "race detection").

* no kernel source writes a bare `s_waitcnt`: every hand-counted wait goes through AMQ_WAIT_VM / AMQ_WAIT_VM_LGKM0 / AMQ_WAIT_LGKM0
  (amq_amd/csrc/amq_common.cuh), so that (a) -DAMQ_WAITS_CONSERVATIVE turns all of them into full drains (the `safe` twin library the GPU
  suite compares the product with, tests/test_gpu_waits.py) and (b) each carries its name and the counts it is derived from into the device
  assembly;
* tools/check_waits.py over the assembly the product objects were assembled from (amq_amd/csrc/asm/*.s, written by the same make rule as the
  objects): per wait, the vector-memory instructions the compiler really emitted between the named points are at least -- and, for the product's
  sites, at their minimum exactly -- what the source counts;
* the checker itself: a synthetic kernel text with a merged load / a moved load / an untagged wait is refused."""
import glob
import os
import re
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "amq_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_waits  # noqa: E402

# the A/B routes (libamq_hip_ab.so: measured negatives, not product) keep their own waits
AB_ONLY = {"amq_engine.hip", "amq_gemv_qkvattn.hip"}


def _product_sources():
    return [p for p in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cuh")) + glob.glob(os.path.join(CSRC, "*.h")))
            if os.path.basename(p) not in AB_ONLY]


def test_no_bare_waitcnt_in_product_sources():
    bad = []
    for p in _product_sources():
        src = open(p).read()
        code = re.sub(r"//[^\n]*", "", src)                          # comments may talk about waits
        code = re.sub(r"/\*.*?\*/", "", code, flags=re.S)
        for m in re.finditer(r"s_waitcnt", code):
            line = code[: m.start()].count("\n") + 1
            ctx = code[code.rfind("\n", 0, m.start()) + 1: code.find("\n", m.start())]
            if os.path.basename(p) == "amq_common.cuh" and ctx.lstrip().startswith("#define AMQ_WAIT"):
                continue
            bad.append(f"{os.path.basename(p)}:{line}: {ctx.strip()[:100]}")
    assert not bad, "bare s_waitcnt (use AMQ_WAIT_VM / AMQ_WAIT_LGKM0, amq_common.cuh):\n" + "\n".join(bad)


def _asm_files():
    d = os.path.join(CSRC, "asm")
    files = sorted(glob.glob(os.path.join(d, "*.s")))
    assert files, "amq_amd/csrc/asm/*.s not found: run __graft_entry__.build() first (the object rule of csrc/Makefile keeps the device assembly)"
    units = {os.path.splitext(os.path.basename(f))[0] for f in files}
    mk = open(os.path.join(CSRC, "Makefile")).read()
    srcs = re.search(r"^SRCS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    missing = [s for s in srcs if os.path.splitext(s)[0] not in units]
    assert not missing, f"no assembly for {missing}: run __graft_entry__.build() first"
    for f in files:                                                   # assembly older than its source: a stale build would check the wrong code
        src = os.path.join(CSRC, os.path.splitext(os.path.basename(f))[0] + ".hip")
        deps = [src, os.path.join(CSRC, "amq_common.cuh"), os.path.join(CSRC, "amq_gemv_body.cuh")]
        assert os.path.getmtime(f) >= max(os.path.getmtime(d_) for d_ in deps if os.path.exists(d_)) - 1.0, \
            f"{os.path.basename(f)} is older than its source: rebuild (make -C amq_amd/csrc)"
    return files


def test_counted_waits_match_the_isa():
    files = _asm_files()
    res = check_waits.run(files=files)
    assert not res["errors"], "\n".join(res["errors"][:20])
    rows = res["rows"]
    by = {}
    for r in rows:
        by.setdefault(r["wait"], []).append(r)
    # every counted site of the product is present and was actually reached from the points it names
    for site in ("gemv.xrows", "f16pp.pro", "f16pp.x", "f16pp.y", "ring.pro", "ring.x0", "ring.x1"):
        assert by.get(site), f"no checked path for wait site {site}"
        assert all(r["ok"] for r in by[site])
    # (the checker's rule is min >= count; the product's sites are additionally pinned to min == count: the waits are as tight as planned)
    for site in ("f16pp.pro", "f16pp.x", "f16pp.y", "ring.pro", "ring.x0", "ring.x1", "ws.pro", "ws.pro2", "ws.w", "ws.p0", "ws.p1"):
        assert all(r["min"] == r["want"] for r in by[site]), site
    # the GEMV row kernels: every body of every (prologue, rows, waves) instantiation stages its rows behind exactly 2 U loads of the ring
    gx = by["gemv.xrows"]
    assert len(gx) >= 100 and all(r["min"] == r["n"] for r in gx if r["file"] != "amq_gemv_pro3.s")
    # (the partial-sum prologue puts its 8 loads of partials and gamma's between the transfers and the ring: stricter than counted, by design)
    assert all(r["min"] > r["n"] for r in gx if r["file"] == "amq_gemv_pro3.s")
    files_with = {r["file"] for r in gx}
    assert files_with == {"amq_gemv_pro0.s", "amq_gemv_pro1.s", "amq_gemv_pro2.s", "amq_gemv_pro3.s"}
    # the ping-pong GEMM: 3 pieces behind X's reads, 5 behind Y's, in both instantiations (fp16, bf16)
    assert {(r["frm"], r["min"], r["max"]) for r in by["f16pp.x"]} == {("f16pp.y", 3, 3), ("f16pp.pro", 3, 3), ("f16pp.drain", 3, 3)}
    assert {(r["min"], r["max"]) for r in by["f16pp.y"]} == {(5, 5)}
    assert res["waits"] >= 150


SYN = """
_Zkernel:
\ts_load_dword s0, s[0:1], 0x0
\tglobal_load_dwordx4 v[0:3], v4, s[2:3]
\t;;#ASMSTART
\tglobal_load_lds_dwordx4 v5, s[2:3] ; AMQ_MARK id=t.dma
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_2
\tglobal_load_dwordx4 v[6:9], v4, s[2:3]
\tglobal_load_dword v10, v4, s[2:3]
\ts_branch .LBB0_3
.LBB0_2:
\tglobal_load_dwordx4 v[6:9], v4, s[2:3]
%s
.LBB0_3:
\t;;#ASMSTART
\ts_waitcnt vmcnt(2) ; AMQ_WAIT id=t.wait n=2 from=t.dma:2
\t;;#ASMEND
%s
\ts_endpgm
.Lfunc_end0:
"""


def _check_text(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(textwrap.dedent(text))
    return check_waits.run(files=[str(p)])


def test_checker_refuses_what_it_should(tmp_path):
    good = _check_text(tmp_path, SYN % ("\tglobal_load_dword v10, v4, s[2:3]", ""))
    assert not good["errors"] and good["rows"][0]["min"] == 2 and good["rows"][0]["max"] == 2
    # one branch issues a single (merged) load: the wait would leave the transfer in flight there
    merged = _check_text(tmp_path, SYN % ("", ""))
    assert any("1..2 vector-memory" in e for e in merged["errors"])
    # one branch issues MORE than counted: stricter than planned, reported in the row, not a violation
    more = _check_text(tmp_path, SYN % ("\tglobal_load_dword v10, v4, s[2:3]\n\tglobal_load_dword v11, v4, s[2:3]", ""))
    assert not more["errors"] and (more["rows"][0]["min"], more["rows"][0]["max"]) == (2, 3)
    # a wait written without the helper
    bare = _check_text(tmp_path, SYN % ("\tglobal_load_dword v10, v4, s[2:3]", "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND"))
    assert any("without an AMQ_WAIT tag" in e for e in bare["errors"])
    # a from-site that does not exist in the kernel
    lost = _check_text(tmp_path, (SYN % ("\tglobal_load_dword v10, v4, s[2:3]", "")).replace("from=t.dma:2", "from=t.nowhere:2"))
    assert any("none of its from-sites" in e for e in lost["errors"])
    # a loop that issues loads between the transfer and the wait: unbounded above, the minimum still decides
    loop = _check_text(tmp_path, (SYN % ("\tglobal_load_dword v10, v4, s[2:3]\n\ts_cbranch_scc0 .LBB0_2", "")))
    assert not loop["errors"] and loop["rows"][0]["max"] is None
