from .agent import Agent
from .ddpg.ddpg_learner import DDPGLearner
from .ddpg.pixel_ddpg_learner import PixelDDPGLearner
from .drq.drq_learner import DrQLearner
from .sac.sac_learner import SACLearner
from .train_state import TempState, TrainState

__all__ = ["Agent", "SACLearner", "DDPGLearner", "PixelDDPGLearner", "DrQLearner", "TrainState", "TempState"]
