"""TEST INFRASTRUCTURE (never on a product path): when this directory is on PYTHONPATH and MORTAL_AMD_TEST_EMU=1, every
`BatchRunner` of the interpreter runs its kernels on the host SIMT emulator (tests/host/emu_pool.py) instead of a GPU — the
only way to execute the reference's UNCHANGED driver scripts (mortal/one_vs_three.py) end to end in the GPU-less container.
tests/test_one_vs_three_script.py puts it there for its subprocess; nothing else does."""
import os
import sys

if os.environ.get("MORTAL_AMD_TEST_EMU") == "1":
    _host = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _root = os.path.dirname(os.path.dirname(_host))
    for _p in (_root, _host):
        if _p not in sys.path:
            sys.path.append(_p)
    import emu_pool

    from mortal_amd import arena as _arena

    _arena.BatchRunner.pool_cls = emu_pool.make_pool_class()
