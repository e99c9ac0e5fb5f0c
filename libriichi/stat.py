"""libriichi.stat (reference libriichi/src/stat.rs): `Stat.from_dir(dir, player_name, disable_progress_bar=False)`,
`Stat.from_log`, 44 counters, derived-rate getters, `total_pt` / `avg_pt` — see mortal_amd/stat.py."""
from mortal_amd.stat import Stat  # noqa: F401

__all__ = ["Stat"]
