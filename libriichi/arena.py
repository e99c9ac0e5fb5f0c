"""libriichi.arena (reference libriichi/src/arena/mod.rs): the HIP-backed batched self-play arena."""
from mortal_amd.arena import OneVsThree, TwoVsTwo  # noqa: F401
