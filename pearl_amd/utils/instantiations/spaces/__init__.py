from .box_action import BoxActionSpace
from .discrete_action import DiscreteActionSpace

__all__ = ["BoxActionSpace", "DiscreteActionSpace"]
