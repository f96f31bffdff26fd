"""The drop-in boundary at the module-path level: the import lines of the reference's training scripts, executed
against `serl_amd.compat.install()`.  The lines are committed here as literal strings (serl_examples/
drq_rlpd_with_vice_reward/binpick_franka_fwbw_vice.py:9-15, drq_rlpd_with_classifier_reward/cable_route_franka_rlpd_
classifier_reward_multithread.py:9-13, ddpg_with_task_reward/pcb_insertion_ddpg.py:9-16, hybrid_pixel_bc/pixel_bc.py:10,16);
env-side imports (serl.evaluation, serl.wrappers, gym, wandb, jax, flax) are not part of the learner path."""
import sys

import pytest

LEARNER_SIDE_IMPORTS = [
    "from serl.agents import SACLearner",
    "from serl.agents import SACLearner, VICELearner",
    "from serl.agents import DrQClassifierRewardLearner",
    "from serl.agents import DrQLearner",
    "from serl.agents import DDPGLearner",
    "from serl.agents import PixelDDPGLearner",
    "from serl.agents import PixelBCLearner, PixelHybridBCLearner",
    "from serl.data import ReplayBuffer, MemoryEfficientReplayBuffer",
    "from serl.data import GPUReplayBuffer",
    "from serl.utils.commons import get_data, restore_checkpoint_, ema",
    "from serl.utils.commons import get_data, restore_checkpoint_",
    "from serl.utils.commons import restore_checkpoint_",
    "from serl.utils.commons import _unpack, _share_encoder, _reset_weights",
    "from serl.utils.augmentations import batched_random_crop, random_crop",
    "from serl.networks import subsample_ensemble",
]


@pytest.fixture
def alias():
    saved = {k: v for k, v in sys.modules.items() if k == "serl" or k.startswith("serl.")}
    for k in saved:
        del sys.modules[k]
    from serl_amd import compat
    yield compat.install(force_stub=True)
    for k in [k for k in sys.modules if k == "serl" or k.startswith("serl.")]:
        del sys.modules[k]
    sys.modules.update(saved)


@pytest.mark.parametrize("line", LEARNER_SIDE_IMPORTS)
def test_reference_import_line(alias, line):
    ns = {}
    exec(line, ns)                                           # noqa: S102  the literal strings above
    for name in line.split(" import ")[1].split(","):
        assert name.strip() in ns


def test_aliases_are_the_serl_amd_objects(alias):
    import serl.agents
    import serl.utils.commons
    import serl_amd.agents
    import serl_amd.utils.commons
    assert serl.agents.SACLearner is serl_amd.agents.SACLearner
    assert serl.utils.commons.ema is serl_amd.utils.commons.ema
    assert "serl.utils.commons" in alias and "serl.distributions" in alias
