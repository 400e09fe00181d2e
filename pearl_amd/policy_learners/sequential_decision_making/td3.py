"""TD3 (reference: pearl/policy_learners/sequential_decision_making/td3.py:43-245).

DDPG plus (a) a delayed actor: the actor step and BOTH target updates only run on rounds where
``_training_steps % actor_update_freq == 0`` (:106-141; the report repeats the last actor loss in
between), with the actor step BEFORE the critic step of the same round, and (b) target policy
smoothing: clipped Gaussian noise, rescaled to the action box, on the target actor's action,
the sum clipped to the box (:143-201).  The noise is the only torch-side random input
(``torch.normal(0, actor_update_noise)`` like the reference, or ``noise_source`` in parity tests);
clamp / rescale / add / clip happen in ``pa_tanh_action``.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import torch
from torch import Tensor

from ...replay_buffers.transition import TransitionBatch
from ..policy_learner import PolicyLearner
from .ddpg import DeepDeterministicPolicyGradient


class TD3(DeepDeterministicPolicyGradient):
    def __init__(self, action_space: Any, *args: Any, actor_update_freq: int = 2,
                 actor_update_noise: float = 0.2, actor_update_noise_clip: float = 0.5,
                 **kwargs: Any) -> None:
        assert hasattr(action_space, "low") and hasattr(action_space, "high"), \
            "TD3 needs a box action space"
        super().__init__(action_space, *args, **kwargs)
        self._actor_update_freq = actor_update_freq
        self._actor_update_noise = actor_update_noise
        self._actor_update_noise_clip = actor_update_noise_clip
        self._last_actor_loss: Any = 0.0
        # parity hook: callable (B, A, device) -> N(0, actor_update_noise^2) draws
        self.noise_source: Optional[Callable[[int, int, torch.device], Tensor]] = None

    def _target_noise(self, B: int, A: int, dev: torch.device):
        if self.noise_source is not None:
            noise = self.noise_source(B, A, dev).to(dev, torch.float32).contiguous()
        else:
            ring = self._flat.get("noise_ring")
            if ring is not None and ring["next"] < ring["buf"].shape[0] and \
                    ring["buf"].shape[1:] == (B, A) and ring["buf"].device == dev:
                noise = ring["buf"][ring["next"]]     # drawn for the whole learn() call at once
                ring["next"] += 1
            else:
                noise = torch.normal(mean=0.0, std=float(self._actor_update_noise), size=(B, A),
                                     device=dev)
        return noise, float(self._actor_update_noise_clip)

    def _begin_learn_loop(self, rounds: int, batch_size: int) -> None:
        """The smoothing noise of every round of this learn() call in ONE generator launch."""
        self._flat.pop("noise_ring", None)
        actor = self._flat.get("actor")
        if self.noise_source is None and rounds > 1 and actor is not None and actor.handle is not None \
                and type(self)._target_noise is TD3._target_noise:
            A = actor.dims[-1]
            if rounds * batch_size * A <= self._NOISE_CHUNK:
                self._flat["noise_ring"] = {"next": 0, "buf": torch.normal(
                    mean=0.0, std=float(self._actor_update_noise), size=(rounds, batch_size, A),
                    device=actor.device)}

    def _end_learn_loop(self) -> None:
        self._flat.pop("noise_ring", None)

    def _learn_batch_device(self, batch: TransitionBatch) -> Dict[str, Any]:
        due = self._training_steps % self._actor_update_freq == 0
        if self._one_call_ok() and type(self)._target_noise is TD3._target_noise:
            actor_loss, critic_loss = self._learn_one_call(batch, due, due)
            if due:
                self._last_actor_loss = actor_loss
            return {"actor_loss": self._last_actor_loss, "critic_loss": critic_loss}
        if due:
            self._last_actor_loss = self._actor_update(batch)
        else:
            self._nets(len(batch))     # the per-batch binding validation _actor_update would do
        self._target_update_follows = due    # then the critics' soft update rides their AdamW launch
        try:
            report = {"actor_loss": self._last_actor_loss, "critic_loss": self._critic_update(batch)}
        finally:
            self._target_update_follows = False
        if due:
            self._update_critic_target()
            self._update_actor_target()
        return report

    # ------------------------------------------------------------------ learn() as one call (ddpg.py)
    def _native_loop_is_mine(self) -> bool:
        cls = type(self)
        return (cls._learn_batch_device is TD3._learn_batch_device
                and cls._learn_one_call is DeepDeterministicPolicyGradient._learn_one_call
                and cls.learn_batch is TD3.learn_batch)

    def _loop_noise(self, rounds: int, B: int, A: int, dev: torch.device):
        if self.noise_source is not None or type(self)._target_noise is not TD3._target_noise:
            return False
        if rounds <= 0:
            return None, float(self._actor_update_noise_clip)
        return (torch.normal(mean=0.0, std=float(self._actor_update_noise), size=(rounds, B, A),
                             device=dev), float(self._actor_update_noise_clip))

    def _loop_report(self, got, step0: int, freq: int) -> Dict[str, Any]:
        # the report repeats the last actor loss on the rounds without an actor step (td3.py:106-141)
        last = self._last_actor_loss
        last = float(last) if isinstance(last, torch.Tensor) else last
        actor_losses = []
        for r, g in enumerate(got[0]):
            if (step0 + r + 1) % freq == 0:
                last = g
            actor_losses.append(last)
        self._last_actor_loss = last
        return {"actor_loss": actor_losses, "critic_loss": got[1]}

    def learn_batch(self, batch: TransitionBatch) -> Dict[str, Any]:
        report = super().learn_batch(batch)
        if isinstance(self._last_actor_loss, torch.Tensor):
            self._last_actor_loss = float(self._last_actor_loss)   # like the reference's .item()
        return report

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, TD3):
            diffs.append("other is not an instance of TD3")
        else:
            for attr in ("_actor_update_freq", "_actor_update_noise", "_actor_update_noise_clip"):
                if getattr(self, attr) != getattr(other, attr):
                    diffs.append(f"{attr} is different: {getattr(self, attr)} vs "
                                 f"{getattr(other, attr)}")
        return "\n".join(d for d in diffs if d)
