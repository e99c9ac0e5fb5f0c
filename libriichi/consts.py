"""libriichi.consts (reference libriichi/src/consts.rs:5-52)."""
MAX_VERSION = 4
ACTION_SPACE = 37 + 1 + 3 + 1 + 1 + 1 + 1 + 1  # discard|kan choice, riichi, chi x3, pon, kan, agari, ryukyoku, pass
GRP_SIZE = 7


def obs_shape(version):
    return {1: (938, 34), 2: (942, 34), 3: (934, 34), 4: (1012, 34)}[version]


def oracle_obs_shape(version):
    return {1: (211, 34), 2: (217, 34), 3: (217, 34), 4: (217, 34)}[version]
