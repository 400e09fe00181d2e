from .q_value_networks import QValueNetwork, VanillaQValueNetwork

__all__ = ["QValueNetwork", "VanillaQValueNetwork"]
