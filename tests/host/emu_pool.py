"""TEST INFRASTRUCTURE: `EmuPool` drives the HOST emulation of the HIP library (tests/host/build_emu.py — the unchanged
kernels of mortal_amd/csrc executed by a fiber-based SIMT emulator) through the same C-ABI and the same `TablePool`
methods as the product path, with CPU tensors instead of HBM.  It exists so that `-m "not gpu"` tests can run the device
code itself against the oracle; nothing in mortal_amd/ imports this module."""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        import build_emu

        from mortal_amd import _lib, tables

        _emu = _lib._load(build_emu.build())
        p = tables.payload()
        if _emu.mj_tables_upload(p, len(p)) != 0:
            raise RuntimeError(_emu.mj_last_error().decode())
    return _emu


def make_pool_class():
    from mortal_amd.pool import TablePool

    class EmuPool(TablePool):
        _L = emu_lib()

        def _stream(self):
            return None

        def _bind_device(self, device):
            self.device = torch.device("cpu")

        def _copy_rows(self, dst, src, nbytes):
            C.memmove(dst, src, nbytes)

    return EmuPool
