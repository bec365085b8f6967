"""CPU oracle for the AMQ mixed-precision dequantize-matmul hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``amq_amd/`` (the product) may import,
call, link or execute anything from this package.  The only permitted users
are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the checker / the timed CPU baseline.

Every function is a from-scratch numpy restatement of the reference algorithm
and cites the reference file:line it follows (paths relative to the reference
checkout, ``amq/kernel/hqq/hqq/...``).

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
real reference (vendored HQQ 0.2.8.post1 under /root/reference) in the build
container with ``tests/golden/gen_golden.py``; ``tests/test_oracle_golden.py``
checks every oracle function bit-exactly (integer work) or to fp16 rounding
(matmul) against those captures.
"""

from . import hqq_ref, gptq_ref, awq_ref, linear_ref  # noqa: F401
