"""Switch the test process to the CPU-emulated host build of the kernels (tools/hipemu) -- TEST INFRASTRUCTURE.

Inside ``with emulated_backend():`` the C-ABI handle cached by ``bbdm_amd._lib`` is the emulator's, and the three
functions through which the package touches the HIP runtime (``_lib.require_gpu / current_stream / device_guard``) are
replaced by CPU no-ops, so the *product* code paths (plans, autograd nodes, the bridge wrappers) run unmodified on CPU
tensors with every kernel executed by the emulator.  The product itself has no such switch: outside this context a CPU
tensor raises ``BBDMHipError`` (tests/test_abi.py::test_product_path_has_no_cpu_fallback)."""
import contextlib
import types

import torch

import kernel_ops as ops
from bbdm_amd import _lib


@contextlib.contextmanager
def emulated_backend():
    emu = ops.use_emulator()
    saved = (_lib._lib, _lib.require_gpu, _lib.current_stream, _lib.device_guard, torch.cuda.synchronize,
             torch.cuda.current_stream)
    _lib._lib = emu
    _lib.require_gpu = lambda *ts: None
    _lib.current_stream = lambda device=None: None
    _lib.device_guard = lambda device=None: contextlib.nullcontext()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=None, synchronize=lambda: None)
    try:
        yield emu
    finally:
        (_lib._lib, _lib.require_gpu, _lib.current_stream, _lib.device_guard, torch.cuda.synchronize,
         torch.cuda.current_stream) = saved
        ops.use_emulator(False)
