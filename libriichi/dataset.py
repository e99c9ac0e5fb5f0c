"""libriichi.dataset (reference libriichi/src/dataset/): `GameplayLoader`, `Gameplay`, `Grp` — see mortal_amd/dataset.py."""
from mortal_amd.dataset import GameplayLoader, Gameplay, Grp  # noqa: F401

__all__ = ["GameplayLoader", "Gameplay", "Grp"]
