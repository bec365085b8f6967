#!/usr/bin/env python3
"""Run a script of this repo against an A/B build of the library instead of the product one.

    python tools/with_variant.py <tag | path/to/lib.so | product> <script.py> [script args ...]

`tag` names amq_amd/libamq_hip_<tag>.so (`make -C amq_amd/csrc variant TAG=<tag> EXTRA=...` / `tuvariant`).  The choice is
made HERE, explicitly, through amq_amd._lib.use_library() before anything loads the library: the product package reads no
environment variable that could swap its numerics."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    tag, script = sys.argv[1], sys.argv[2]
    from amq_amd import _lib
    if tag not in ("", "product"):
        path = tag if tag.endswith(".so") else os.path.join(ROOT, "amq_amd", f"libamq_hip_{tag}.so")
        if not os.path.exists(path):
            raise SystemExit(f"{path} not found (make -C amq_amd/csrc variant TAG={tag} EXTRA=...)")
        _lib.use_library(path)
    sys.argv = [script] + sys.argv[3:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
