from .neural_linear_bandit import NeuralLinearBandit, SquareCBExploration

__all__ = ["NeuralLinearBandit", "SquareCBExploration"]
