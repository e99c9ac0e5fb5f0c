"""libriichi.mjai (reference libriichi/src/mjai/): `Bot(engine, player_id).react(line, can_act=True)` — see
mortal_amd/mjai.py."""
from mortal_amd.mjai import Bot  # noqa: F401

__all__ = ["Bot"]
