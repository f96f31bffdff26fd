from .vice_learner import VICELearner

__all__ = ["VICELearner"]
