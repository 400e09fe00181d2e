from .basic_replay_buffer import BasicReplayBuffer, TensorBasedReplayBuffer
from .bootstrap_replay_buffer import BootstrapReplayBuffer
from .hindsight_experience_replay_buffer import HindsightExperienceReplayBuffer
from .replay_buffer import ReplayBuffer
from .sarsa_replay_buffer import SARSAReplayBuffer
from .transition import (Transition, TransitionBatch, TransitionWithBootstrapMask,
                         TransitionWithBootstrapMaskBatch, filter_batch_by_bootstrap_mask)

__all__ = ["BasicReplayBuffer", "BootstrapReplayBuffer", "HindsightExperienceReplayBuffer",
           "TensorBasedReplayBuffer", "ReplayBuffer", "SARSAReplayBuffer", "Transition",
           "TransitionBatch", "TransitionWithBootstrapMask", "TransitionWithBootstrapMaskBatch",
           "filter_batch_by_bootstrap_mask"]
