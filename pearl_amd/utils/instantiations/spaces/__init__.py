from .discrete_action import DiscreteActionSpace

__all__ = ["DiscreteActionSpace"]
