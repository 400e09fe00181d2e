from .policy_learner import PolicyLearner

__all__ = ["PolicyLearner"]
