from .agent import Agent
from .bc.pixel_bc_learner import PixelBCLearner
from .bc.pixel_hybrid_bc_learner import PixelHybridBCLearner
from .ddpg.ddpg_learner import DDPGLearner
from .ddpg.pixel_ddpg_learner import PixelDDPGLearner
from .drq.drq_classifier_reward_learner import DrQClassifierRewardLearner
from .drq.drq_learner import DrQLearner
from .sac.sac_learner import SACLearner
from .train_state import TempState, TrainState
from .vice.vice_learner import VICELearner

__all__ = ["Agent", "PixelBCLearner", "PixelHybridBCLearner", "SACLearner", "DDPGLearner", "PixelDDPGLearner", "DrQLearner", "DrQClassifierRewardLearner", "VICELearner", "TrainState", "TempState"]
