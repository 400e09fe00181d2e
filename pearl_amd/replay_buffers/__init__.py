from .basic_replay_buffer import BasicReplayBuffer, TensorBasedReplayBuffer
from .hindsight_experience_replay_buffer import HindsightExperienceReplayBuffer
from .replay_buffer import ReplayBuffer
from .sarsa_replay_buffer import SARSAReplayBuffer
from .transition import Transition, TransitionBatch

__all__ = ["BasicReplayBuffer", "HindsightExperienceReplayBuffer", "TensorBasedReplayBuffer", "ReplayBuffer", "SARSAReplayBuffer", "Transition",
           "TransitionBatch"]
