"""torch_random_walk_amd -- MI355X-native CSR random walks (uniform + node2vec).

One hot path of Sachinx0e/torch_random_walk, rebuilt for gfx950 behind the
reference's own surface::

    from torch_rw import rw, utils                      # drop-in names
    walks = rw.walk(row_ptr=..., col_idx=..., target_nodes=..., p=1.0, q=1.0,
                    walk_length=80, seed=10)             # [N, 81] int64 on the GPU

The compute lives in ``libtrw_hip.so`` (hand-written HIP, C ABI in
``include/trw.h``); ``torch_rw_native`` is the pybind host extension that
mirrors the reference's native module and its dispatcher: GPU tensors run the HIP
kernels, CPU tensors the library's own host legs (``csrc/trw_cpu.cpp``, what the
reference's ``csrc/cpu`` computes on one thread).  Nothing falls back silently:
importing this package without the built extension raises, and a GPU call never
runs on the host.
"""
from . import _native  # noqa: F401  (raises if the HIP extension is not built)
from . import rw, utils  # noqa: F401
from ._native import RNG_AUTO, RNG_GLIBC, RNG_WALKER  # noqa: F401

__version__ = "0.1.0"
