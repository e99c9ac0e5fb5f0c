"""libriichi.mjai — a "next" row of the hot-path scope table; not built this round."""


def __getattr__(name):
    raise NotImplementedError(f"libriichi.mjai.{name} is not implemented yet (SURVEY.md §8(f))")
