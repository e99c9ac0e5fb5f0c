"""mortal_amd — MI355X-native batched riichi-mahjong arena (hot path of Equim-chan/Mortal's `libriichi`).

Import is cheap; the HIP library is loaded on first use of `mortal_amd.pool` / `mortal_amd.arena`.
"""
__version__ = "0.1.0"
