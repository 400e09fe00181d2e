from .neural_linear_bandit import NeuralLinearBandit, SquareCBExploration, UCBExploration

__all__ = ["NeuralLinearBandit", "SquareCBExploration", "UCBExploration"]
