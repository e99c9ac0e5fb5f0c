"""libriichi.state (reference libriichi/src/state/): `PlayerState(player_id)` with `update(json) -> ActionCandidate`,
`encode_obs(version, at_kan_select)`, getters, `brief_info()` — on the device path, see mortal_amd/state.py."""
from mortal_amd.state import ActionCandidate, PlayerState  # noqa: F401

__all__ = ["PlayerState", "ActionCandidate"]
