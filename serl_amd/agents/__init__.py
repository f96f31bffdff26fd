from .agent import Agent
from .ddpg.ddpg_learner import DDPGLearner
from .drq.drq_learner import DrQLearner
from .sac.sac_learner import SACLearner
from .train_state import TempState, TrainState

__all__ = ["Agent", "SACLearner", "DDPGLearner", "DrQLearner", "TrainState", "TempState"]
