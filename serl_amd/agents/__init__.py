from .agent import Agent
from .ddpg.ddpg_learner import DDPGLearner
from .ddpg.pixel_ddpg_learner import PixelDDPGLearner
from .drq.drq_classifier_reward_learner import DrQClassifierRewardLearner
from .drq.drq_learner import DrQLearner
from .sac.sac_learner import SACLearner
from .train_state import TempState, TrainState
from .vice.vice_learner import VICELearner

__all__ = ["Agent", "SACLearner", "DDPGLearner", "PixelDDPGLearner", "DrQLearner", "DrQClassifierRewardLearner", "VICELearner", "TrainState", "TempState"]
