"""Shim: the reference's mortal/prelude.py:23 does `import torch.utils.tensorboard` (only to pre-empt a deprecation warning), and
mortal/train.py / train_grp.py build a `SummaryWriter`; this image has no `tensorboard` wheel.  With `compat/` on PYTHONPATH this
package satisfies torch's import (`torch/utils/tensorboard/__init__.py` checks `tensorboard.__version__ >= 1.15`, then pulls
protobuf / writer classes out of a dozen `tensorboard.*` sub-modules): every sub-module resolves to a stub whose attributes are
inert classes, so `torch.utils.tensorboard.SummaryWriter(...)` constructs and every `add_*` / `flush` / `close` call is a no-op.
Nothing is written anywhere.

INTEGRATION.md puts `compat/` FIRST on PYTHONPATH (the toml shim needs that), so this package would shadow a real tensorboard
on a machine that has one and train.py's SummaryWriter would silently write nothing.  Therefore: at import the rest of sys.path
is searched for a real `tensorboard`; if there is one it is loaded in this package's place (`sys.modules["tensorboard"]` becomes
the real package), otherwise the no-op shim installs itself and says so once on stderr."""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types


def _real_tensorboard():
    """The spec of a `tensorboard` package found on sys.path outside this directory's parent, or None."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    others = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != here]
    try:
        return importlib.machinery.PathFinder.find_spec("tensorboard", others)
    except (ImportError, ValueError):
        return None


_spec = _real_tensorboard()
if _spec is not None and _spec.loader is not None:
    _real = importlib.util.module_from_spec(_spec)
    sys.modules[__name__] = _real  # `import tensorboard` (this very import) hands back the real package
    _spec.loader.exec_module(_real)
else:
    print("mortal_amd compat: no tensorboard installed, SummaryWriter calls are no-ops (compat/tensorboard shim)", file=sys.stderr)

__version__ = "2.15.0+mortal-amd-noop-shim"


class _Inert:
    """Accepts any construction, call, attribute or item access and does nothing (protobuf messages, writers, enum values)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Inert()

    def __setattr__(self, name, value):
        pass

    def __iter__(self):
        return iter(())

    def __len__(self):
        return 0

    def __bool__(self):
        return False

    def __getitem__(self, key):
        return _Inert()

    def __setitem__(self, key, value):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _InertMeta(type):
    def __getattr__(cls, name):  # nested message types / enum constants: Summary.Value, SessionLog.START ...
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _stub_class(name)


def _stub_class(name):
    return _InertMeta(name, (_Inert,), {})


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        value = _stub_class(name)
        setattr(self, name, value)
        return value


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(__name__ + "."):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _StubModule(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        if module.__name__ == __name__ + ".summary.writer.event_file_writer":
            module.EventFileWriter = _NullEventFileWriter


class _NullEventFileWriter:
    """What torch's FileWriter drives (tensorboard/summary/writer/event_file_writer.py): remembers the directory, drops every event."""

    def __init__(self, logdir, max_queue_size=10, flush_secs=120, filename_suffix=""):
        self._logdir = str(logdir)

    def get_logdir(self):
        return self._logdir

    def add_event(self, event):
        pass

    def flush(self):
        pass

    def close(self):
        pass


if _spec is None and not any(isinstance(f, _Finder) for f in sys.meta_path):  # (never next to a real tensorboard)
    sys.meta_path.append(_Finder())
