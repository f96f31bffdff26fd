from .ddpg_learner import DDPGLearner
from .pixel_ddpg_learner import PixelDDPGLearner
