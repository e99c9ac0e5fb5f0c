"""Drop-in for the public surface of Mortal's `libriichi` module (reference libriichi/src/lib.rs:137-160) backed by
the MI355X table pool (mortal_amd).  Put the repository root on PYTHONPATH and the reference's drivers
(`mortal/one_vs_three.py`, `mortal/player.py`) import this package unchanged.

Implemented: `libriichi.consts`, `libriichi.arena` (OneVsThree/TwoVsTwo `py_vs_py`, incl. `log_dir` mjai dumps),
`libriichi.stat.Stat`, `libriichi.dataset` (GameplayLoader incl. oracle=True, Gameplay, Grp), `libriichi.state.PlayerState`
(update / validate_reaction / encode_obs / getters), `libriichi.mjai.Bot`.
"""
import importlib as _il
import sys as _sys

__profile__ = "release"
__version__ = "0.1.0+mortal_amd"

from . import consts  # noqa: E402,F401


def __getattr__(name):
    if name in ("arena", "stat", "dataset", "mjai", "state"):
        mod = _il.import_module(f"{__name__}.{name}")
        setattr(_sys.modules[__name__], name, mod)
        return mod
    raise AttributeError(name)
