"""xeve_amd -- MI355X (gfx950) implementation of XEVE's inter-prediction / RDO arithmetic hot path.

The product is the C-ABI shared library ``xeve_amd/lib/libxeve_hip.so`` (include/xeve_hip.h): hand-written
HIP kernels behind (1) drop-in dispatch tables with the reference's exact function-pointer signatures and
(2) a batched device API.  This Python package is only the thin host-side plumbing the tests, bench.py and
the GOP-shard driver need: a ctypes binding (``lib``), a mirror of the reference's dispatch-table interface
(``tables``), torch-tensor wrappers of the batched API (``device``) and the closed-GOP shard planner (``gop``).

There is no CPU fallback anywhere in this package: if the HIP library is missing or no gfx950 device is
usable, loading / initialising raises.
"""
from .lib import XeveHipError, init, last_error, load, table_calls  # noqa: F401
