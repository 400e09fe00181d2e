from .deep_q_learning import DeepQLearning

__all__ = ["DeepQLearning"]
