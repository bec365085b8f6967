#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every gfx950 kernel in a hipcc object (or .so): tools/kernel_regs.py amq_amd/csrc/amq_gemv.o [substring]
(extracts .hip_fatbin, unbundles the gfx950 code object, reads its metadata notes; no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(path):
    with tempfile.TemporaryDirectory() as d:
        fb, co = os.path.join(d, "fb"), os.path.join(d, "co")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fb, "--output=" + co,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for k in re.split(r"- \.agpr_count", notes)[1:]:
        g = lambda key: re.search(r"\." + key + r":\s+(\S+)", k).group(1)
        out.append(dict(name=g("name"), vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), scratch=int(g("private_segment_fixed_size")),
                        lds=int(g("group_segment_fixed_size")), agpr=int(re.match(r":\s+(\d+)", k).group(1))))
    return out


if __name__ == "__main__":
    ks = kernels(sys.argv[1])
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in ks), capture_output=True, text=True).stdout.split("\n")
    for k, n in zip(ks, names):
        n = re.sub(r"\(.*", "", n.replace("void amq::", ""))
        if len(sys.argv) > 2 and sys.argv[2] not in n:
            continue
        print("%4d vgpr %3d agpr %3d sgpr %5d scratch %6d lds  %s" % (k["vgpr"], k["agpr"], k["sgpr"], k["scratch"], k["lds"], n))
