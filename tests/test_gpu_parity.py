"""-m gpu: the HIP table pool vs the oracle, bit for bit, through the C-ABI (BASELINE configs[1]: random-action
policy, env-step kernels; state transitions, masks, obs and scores must be identical)."""
import pytest

import parity_util

pytestmark = pytest.mark.gpu


def test_lockstep_v3_256_tables(oracle):
    """256 tables, full hanchan, every cycle: row lists, 46-wide masks and the whole v3 obs (934x34 f32) bit-exact."""
    st = parity_util.run_lockstep(oracle, 256, version=3, max_cycles=3000, obs_every=1)
    assert st["scores_checked"] == 256 and st["obs_checked"] > 100000


def test_lockstep_v4_nonsp_rows(oracle):
    """v4 layout: rows 0..888 (everything except the SP block) bit-exact; masks every cycle."""
    st = parity_util.run_lockstep(oracle, 64, version=4, max_cycles=3000, obs_every=7)
    assert st["scores_checked"] == 64


def test_lockstep_4096_tables_masks_scores(oracle):
    """BASELINE configs[1] size: 4096 tables, masks + row lists every cycle, final scores of every game."""
    st = parity_util.run_lockstep(oracle, 4096, version=3, max_cycles=3000, compare_obs=False)
    assert st["scores_checked"] == 4096
    assert st["counters"]["steps"] == st["oracle_steps"]
