"""BASELINE configs[0] literally: the reference's `mortal/one_vs_three.py` is launched UNCHANGED as a subprocess —
`import prelude` (which imports torch.utils.tensorboard: compat/tensorboard), `from config import config` (compat/toml),
`from libriichi.arena import OneVsThree` (this repository's package) — with a generated config.toml and two torch.save'd tiny
random-init checkpoints, 8 hanchan (games_per_iter = 8), and its printed line `challenger rankings: [...]` is compared with the
reference-shaped oracle loop driven by the same two networks.  This is INTEGRATION.md §1's recipe, verbatim, except that the
subprocess has no GPU: a test-side sitecustomize (tests/host/sitecustomize_emu) points the arena at the host emulator build of
the same kernels.  Skipped where /root/reference does not exist (the GPU box).

Round 6: the same test against the REAL library on an MI355X — `tools/r06_cfg0_gpu.sh` ships a scratch copy of
the reference's mortal/ directory with one builder `gpurun` call (git-ignored, deleted afterwards), sets MORTAL_REF_DIR to it and
MORTAL_AMD_CFG0_REAL=1: config devices 'cuda:0', no emulator injection, libmortal_amd.so loaded by the unchanged script; the oracle
loop's two engines run on the same GPU.  The log of that run is kept under profiles/."""
import os
import re
import subprocess
import sys

import pytest

REF = os.environ.get("MORTAL_REF_DIR", "/root/reference/mortal")
REAL = os.environ.get("MORTAL_AMD_CFG0_REAL") == "1"  # the real libmortal_amd.so on cuda:0 instead of the host emulator
DEVICE = "cuda:0" if REAL else "cpu"
KEY = 0x55DFAA4CEF265CD7  # a TOML integer is a signed 64-bit value (config.toml: seed_key)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "one_vs_three.py")), reason="/root/reference not present")

VERSION = 4


def _save_checkpoint(ref_model, path, seed):
    """What mortal/one_vs_three.py:27-35 reads: config.control.version, config.resnet.{conv_channels,num_blocks}, mortal, current_dqn."""
    import torch

    torch.manual_seed(seed)
    brain = ref_model.Brain(version=VERSION, conv_channels=16, num_blocks=1)
    dqn = ref_model.DQN(version=VERSION)
    torch.save({"config": {"control": {"version": VERSION}, "resnet": {"conv_channels": 16, "num_blocks": 1}},
                "mortal": brain.state_dict(), "current_dqn": dqn.state_dict()}, path)


def test_one_vs_three_script_runs_unchanged(oracle, tmp_path):
    for p in (ROOT, os.path.join(ROOT, "compat"), HOST, REF):
        if p not in sys.path:
            sys.path.append(p)
    import engine as ref_engine  # mortal/engine.py
    import model as ref_model  # mortal/model.py
    import torch

    import test_reference_engine as T
    from mortal_amd.pool import default_deal_algo

    cham_pt, chal_pt = str(tmp_path / "champion.pth"), str(tmp_path / "challenger.pth")
    _save_checkpoint(ref_model, cham_pt, 2)
    _save_checkpoint(ref_model, chal_pt, 1)
    cfg = tmp_path / "config.toml"
    cfg.write_text(f"""
[1v3]
seed_key = {KEY}
games_per_iter = 8
iters = 1
log_dir = '{tmp_path / "logs"}'

[1v3.champion]
device = '{DEVICE}'
enable_compile = false
enable_amp = false
enable_rule_based_agari_guard = false
name = 'champion'
state_file = '{cham_pt}'

[1v3.challenger]
device = '{DEVICE}'
enable_compile = false
enable_amp = false
enable_rule_based_agari_guard = false
name = 'challenger'
state_file = '{chal_pt}'

[1v3.akochan]
enabled = false
""")
    # repo root and compat/ before anything else on PYTHONPATH; + the test-only emulator injection
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "compat")] + ([] if REAL else [os.path.join(HOST, "sitecustomize_emu")])),
               MORTAL_CFG=str(cfg), OMP_NUM_THREADS="4")
    if REAL:
        env.pop("MORTAL_AMD_TEST_EMU", None)
    else:
        env["MORTAL_AMD_TEST_EMU"] = "1"
    run = subprocess.run([sys.executable, os.path.join(REF, "one_vs_three.py")], cwd=REF, env=env, stdin=subprocess.DEVNULL,
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    m = re.search(r"challenger rankings: \[\s*(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s*\]", run.stdout)
    assert m, run.stdout[-2000:]
    got = [int(x) for x in m.groups()]
    assert sum(got) == 8

    # the same two checkpoints through the reference's MortalEngine on the oracle loop (arena/game.rs:286-304)
    def load(path, name):
        st = torch.load(path, weights_only=True, map_location="cpu")
        brain = ref_model.Brain(version=VERSION, conv_channels=16, num_blocks=1).eval().to(DEVICE)
        dqn = ref_model.DQN(version=VERSION).eval().to(DEVICE)
        brain.load_state_dict(st["mortal"])
        dqn.load_state_dict(st["current_dqn"])
        return ref_engine.MortalEngine(brain, dqn, is_oracle=False, version=VERSION, device=torch.device(DEVICE), enable_amp=False,
                                       enable_rule_based_agari_guard=False, name=name)

    want = T._oracle_rankings(oracle, load(chal_pt, "challenger"), load(cham_pt, "champion"), (10000, KEY), 2, VERSION, default_deal_algo())
    assert got == want, (got, want, run.stdout[-500:])
    print(f"cfg0 ({'libmortal_amd.so on ' + DEVICE if REAL else 'host emulator'}): challenger rankings {got} == oracle loop {want}; script stdout tail: {run.stdout[-300:]!r}")
    # log_dir side effect (one_vs_three.rs:127-129,195-225): one gzip'd mjai log per hanchan
    logs = sorted(os.listdir(tmp_path / "logs"))
    assert len(logs) == 8 and all(f.endswith(".json.gz") for f in logs), logs
