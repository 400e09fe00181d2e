"""CPU restatement of the reference's actor-critic learner paths (PPO, continuous SAC).

TEST INFRASTRUCTURE ONLY — the checker, never the product.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file; pearl_amd/ must not (and does not).

Plain PyTorch fp32 on the CPU; forward formulas are written out, gradients come from torch
autograd and the optimizer is ``torch.optim.AdamW(amsgrad=True)`` itself — that IS the reference's
arithmetic (ATen), there is nothing lower to restate.  What each piece follows (file:line under
/root/reference):

  mlp                       pearl/neural_networks/common/utils.py:75-152 (Linear+ReLU ..., Linear)
  PpoOracle.action_probs    neural_networks/sequential_decision_making/actor_networks.py:155-176
  PpoOracle.preprocess      policy_learners/sequential_decision_making/ppo.py:201-293 (GAE, lambda return)
  PpoOracle.actor_loss      ppo.py:152-183     PpoOracle.critic_loss  ppo.py:185-192 + critic_utils.py:139-167
  PpoOracle.learn           ppo.py:194-199 + policy_learner.py:162-195 + actor_critic_base.py:309-366
  SacOracle.sample_action   actor_networks.py:537-591
  SacOracle.actor_loss      soft_actor_critic_continuous.py:208-231
  SacOracle.critic_loss     :155-206 + critic_utils.py:170-203
  SacOracle.learn_batch     actor_critic_base.py:309-366 + soft_actor_critic_continuous.py:131-153

Parity is PINNED: tests/test_oracle_ac_golden.py checks these against fixtures minted by running
the real reference (oracle/make_golden_ac.py -> tests/golden/ppo_*.pt, sac_*.pt).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
from torch import Tensor

F32 = torch.float32


def _layers(sd: Dict[str, Tensor], prefix: str = "_model.") -> List[Tuple[Tensor, Tensor]]:
    """(weight, bias) leaf tensors of an mlp_block state dict, in layer order."""
    out, i = [], 0
    while f"{prefix}{i}.0.weight" in sd:
        out.append((sd[f"{prefix}{i}.0.weight"].clone().requires_grad_(True),
                    sd[f"{prefix}{i}.0.bias"].clone().requires_grad_(True)))
        i += 1
    return out


def mlp(layers: Sequence[Tuple[Tensor, Tensor]], x: Tensor, last_relu: bool = False) -> Tensor:
    for i, (w, b) in enumerate(layers):
        x = torch.nn.functional.linear(x, w, b)
        if i + 1 < len(layers) or last_relu:
            x = torch.relu(x)
    return x


def _adamw(params: List[Tensor], lr: float) -> torch.optim.Optimizer:
    return torch.optim.AdamW([{"params": params, "lr": lr, "amsgrad": True}])


def _flat(layers) -> List[Tensor]:
    return [t for wb in layers for t in wb]


# --------------------------------------------------------------------------------------
# PPO
# --------------------------------------------------------------------------------------
class PpoOracle:
    def __init__(self, actor_sd, critic_sd, A: int, gamma: float = 0.99, lam: float = 0.95,
                 epsilon: float = 0.0, entropy_scale: float = 0.01, lr: float = 1e-4) -> None:
        self.actor, self.critic = _layers(actor_sd), _layers(critic_sd)
        self.A, self.gamma, self.lam, self.eps, self.ent = A, gamma, lam, epsilon, entropy_scale
        self.opt_a, self.opt_c = _adamw(_flat(self.actor), lr), _adamw(_flat(self.critic), lr)

    def action_probs(self, state: Tensor, onehot: Tensor) -> Tensor:
        probs = torch.softmax(mlp(self.actor, state), dim=-1)
        return torch.sum(probs * onehot, dim=1, keepdim=True).view(-1)

    @torch.no_grad()
    def preprocess(self, states: Tensor, onehot: Tensor, reward: Tensor, term: Tensor,
                   trunc: Tensor, last_next_state: Tensor):
        """Logical order in (index 0 = oldest), logical order out."""
        n = states.shape[0]
        values = mlp(self.critic, states).view(-1)
        aprob = self.action_probs(states, onehot)
        next_value = mlp(self.critic, last_next_state.view(1, -1)).view(-1)[0]
        gae = torch.tensor(0.0)
        out_g, out_r = torch.empty(n), torch.empty(n)
        for i in range(n - 1, -1, -1):
            td = reward[i] + self.gamma * next_value * (~term[i]) - values[i]
            gae = td + self.gamma * self.lam * (not (bool(term[i]) or bool(trunc[i]))) * gae
            out_g[i], out_r[i] = gae, gae + values[i]
            next_value = values[i]
        return out_g, out_r, aprob

    def actor_loss(self, state, onehot, p_old, gae) -> Tensor:
        p = self.action_probs(state, onehot)
        r = torch.div(p, p_old)
        clip = torch.clamp(r, min=1.0 - self.eps, max=1.0 + self.eps)
        loss = torch.sum(-torch.min(r * gae, clip * gae))
        entropy = torch.distributions.Categorical(p.detach()).entropy()
        return loss - torch.sum(self.ent * entropy)

    def critic_loss(self, state, lam_return) -> Tensor:
        vs = mlp(self.critic, state)
        return torch.nn.MSELoss()(vs.reshape_as(lam_return), lam_return.detach())

    def learn_batch(self, state, onehot, p_old, gae, lam_return) -> Tuple[float, float]:
        la = self.actor_loss(state, onehot, p_old, gae)
        self.opt_a.zero_grad()
        la.backward()
        self.opt_a.step()
        self.opt_c.zero_grad()
        lc = self.critic_loss(state, lam_return)
        lc.backward()
        self.opt_c.step()
        return la.item(), lc.item()


# --------------------------------------------------------------------------------------
# continuous SAC
# --------------------------------------------------------------------------------------
class SacOracle:
    def __init__(self, actor_sd, critic_sd, critic_target_sd, low: Tensor, high: Tensor,
                 gamma: float = 0.99, tau: float = 0.005, lr: float = 1e-3) -> None:
        self.trunk = _layers(actor_sd)
        self.head = [actor_sd[k].clone().requires_grad_(True) for k in
                     ("fc_mu.weight", "fc_mu.bias", "fc_std.weight", "fc_std.bias")]
        self.c = [_layers(critic_sd, f"_critic_{i}._model.") for i in (1, 2)]
        self.ct = [[(w.detach().clone(), b.detach().clone()) for w, b in
                    _layers(critic_target_sd, f"_critic_{i}._model.")] for i in (1, 2)]
        self.low, self.high = low, high
        self.bound = (high - low) / 2
        self.gamma, self.tau = gamma, tau
        self.log_alpha = torch.zeros(1, requires_grad=True)
        self.alpha = torch.exp(self.log_alpha).detach()
        self.target_entropy = -torch.tensor(low.shape[0])
        self.opt_a = _adamw(_flat(self.trunk) + self.head, lr)
        self.opt_c = _adamw(_flat(self.c[0]) + _flat(self.c[1]), lr)
        self.opt_e = torch.optim.AdamW([self.log_alpha], lr=lr, amsgrad=True)

    def sample_action(self, state: Tensor, noise: Tensor) -> Tuple[Tensor, Tensor]:
        x = mlp(self.trunk, state, last_relu=True)
        mean = torch.nn.functional.linear(x, self.head[0], self.head[1])
        log_std = torch.tanh(torch.nn.functional.linear(x, self.head[2], self.head[3]))
        log_std = -5 + 0.5 * (2 - (-5)) * (log_std + 1)
        std = log_std.exp()
        sample = mean + noise * std                                   # Normal.rsample
        n = torch.tanh(sample)
        action = (((self.high - self.low) * (n + 1.0)) / 2) + self.low
        var = std ** 2
        log_prob = -((sample - mean) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi))
        log_prob = log_prob - torch.log(self.bound * (1 - n.pow(2)) + 1e-6)
        return action, log_prob.sum(dim=1, keepdim=True)

    @staticmethod
    def q(layers, state, action) -> Tensor:
        return mlp(layers, torch.cat([state, action], dim=-1)).view(-1)

    def learn_batch(self, batch: Dict[str, Tensor], noise_actor: Tensor, noise_critic: Tensor
                    ) -> Dict[str, float]:
        s, a, r, term, ns = (batch[k] for k in ("state", "action", "reward", "terminated", "next_state"))
        # ---- actor (:208-231)
        act, logp = self.sample_action(s, noise_actor)
        qmin = torch.minimum(self.q(self.c[0], s, act), self.q(self.c[1], s, act)).unsqueeze(-1)
        actor_loss = (self.alpha * logp - qmin).mean()
        self.opt_a.zero_grad()
        actor_loss.backward()
        self.opt_a.step()
        # ---- critic (:155-206)
        self.opt_c.zero_grad()
        with torch.no_grad():
            nact, nlogp = self.sample_action(ns, noise_critic)
            nq = torch.minimum(self.q(self.ct[0], ns, nact), self.q(self.ct[1], ns, nact)).unsqueeze(-1)
            nv = (nq - self.alpha * nlogp).view(-1)
            y = (nv * self.gamma * (1 - term.float())) + r
        mse = torch.nn.MSELoss()
        critic_loss = (mse(self.q(self.c[0], s, a), y) + mse(self.q(self.c[1], s, a), y)) / 2.0
        critic_loss.backward()
        self.opt_c.step()
        # ---- targets (critic_utils.py:103-136)
        with torch.no_grad():
            for net, tgt in zip(self.c, self.ct):
                for (w, b), (tw, tb) in zip(net, tgt):
                    tw.copy_(self.tau * w + (1.0 - self.tau) * tw)
                    tb.copy_(self.tau * b + (1.0 - self.tau) * tb)
        # ---- entropy autotune (:134-151)
        ent_loss = (-torch.exp(self.log_alpha) * (logp + self.target_entropy).detach()).mean()
        self.opt_e.zero_grad()
        ent_loss.backward()
        self.opt_e.step()
        self.alpha = torch.exp(self.log_alpha).detach()
        return {"actor_loss": actor_loss.item(), "critic_loss": critic_loss.item(),
                "entropy_coef": ent_loss.item()}


# --------------------------------------------------------------------------------------
# discrete SAC
# --------------------------------------------------------------------------------------
class DiscreteSacOracle:
    """SoftActorCritic (soft_actor_critic.py:46-330): softmax actor, twin critics on every available
    action, expected-value target, entropy autotune with Adam(eps=1e-4)."""

    def __init__(self, actor_sd, critic_sd, critic_target_sd, n_actions: int, gamma: float = 0.99,
                 tau: float = 0.005, lr: float = 1e-4, target_entropy_scale: float = 0.89) -> None:
        self.actor = _layers(actor_sd)
        self.c = [_layers(critic_sd, f"_critic_{i}._model.") for i in (1, 2)]
        self.ct = [[(w.detach().clone(), b.detach().clone()) for w, b in
                    _layers(critic_target_sd, f"_critic_{i}._model.")] for i in (1, 2)]
        self.gamma, self.tau = gamma, tau
        self.log_alpha = torch.zeros(1, requires_grad=True)
        self.alpha = torch.exp(self.log_alpha).detach()
        self.target_entropy = -target_entropy_scale * torch.log(1.0 / torch.tensor(n_actions))
        self.opt_a = _adamw(_flat(self.actor), lr)
        self.opt_c = _adamw(_flat(self.c[0]) + _flat(self.c[1]), lr)
        self.opt_e = torch.optim.Adam([self.log_alpha], lr=lr, eps=1e-4)

    def policy(self, state: Tensor) -> Tensor:
        return torch.softmax(mlp(self.actor, state), dim=-1)

    @staticmethod
    def q_all(layers, state: Tensor, reps: Tensor) -> Tensor:
        """Q(s_b, a_i) for every available-action slot: (B, A)."""
        B, A, _ = reps.shape
        s = state.unsqueeze(1).expand(B, A, state.shape[1])
        return mlp(layers, torch.cat([s, reps], dim=-1)).view(B, A)

    @staticmethod
    def q(layers, state, action) -> Tensor:
        return mlp(layers, torch.cat([state, action], dim=-1)).view(-1)

    def learn_batch(self, batch: Dict[str, Tensor]) -> Dict[str, float]:
        s, a, r, term, ns = (batch[k] for k in ("state", "action", "reward", "terminated", "next_state"))
        # ---- actor (:254-287)
        qmin = torch.minimum(self.q_all(self.c[0], s, batch["curr_available_actions"]),
                             self.q_all(self.c[1], s, batch["curr_available_actions"])).detach().clone()
        if batch.get("curr_unavailable_actions_mask") is not None:
            qmin[batch["curr_unavailable_actions_mask"]] = 0.0
        p = self.policy(s)
        logp = torch.log(p + 1e-8)
        actor_loss = (p * (self.alpha * logp - qmin)).mean()
        self.opt_a.zero_grad()
        actor_loss.backward()
        self.opt_a.step()
        p_cache, logp_cache = p.detach(), logp.detach()
        # ---- critic (:153-252), with the updated actor
        self.opt_c.zero_grad()
        with torch.no_grad():
            nq = torch.minimum(self.q_all(self.ct[0], ns, batch["next_available_actions"]),
                               self.q_all(self.ct[1], ns, batch["next_available_actions"]))
            if batch.get("next_unavailable_actions_mask") is not None:
                nq[batch["next_unavailable_actions_mask"]] = 0.0
            npi = self.policy(ns)
            nv = ((nq - self.alpha * torch.log(npi + 1e-8)) * npi).sum(dim=1)
            y = (nv * self.gamma * (1 - term.float())) + r
        mse = torch.nn.MSELoss()
        critic_loss = (mse(self.q(self.c[0], s, a), y) + mse(self.q(self.c[1], s, a), y)) / 2.0
        critic_loss.backward()
        self.opt_c.step()
        with torch.no_grad():
            for net, tgt in zip(self.c, self.ct):
                for (w, b), (tw, tb) in zip(net, tgt):
                    tw.copy_(self.tau * w + (1.0 - self.tau) * tw)
                    tb.copy_(self.tau * b + (1.0 - self.tau) * tb)
        # ---- entropy autotune (:134-151)
        entropy = -(p_cache * logp_cache).sum(1).mean()
        ent_loss = torch.exp(self.log_alpha) * (entropy - self.target_entropy).detach()
        self.opt_e.zero_grad()
        ent_loss.backward()
        self.opt_e.step()
        self.alpha = torch.exp(self.log_alpha).detach()
        return {"actor_loss": actor_loss.item(), "critic_loss": critic_loss.item(),
                "entropy_coef": ent_loss.item()}


# --------------------------------------------------------------------------------------
# implicit Q-learning
# --------------------------------------------------------------------------------------
class IqlOracle:
    """ImplicitQLearning (implicit_q_learning.py:159-285) with a deterministic tanh actor
    (``continuous=True``: weighted MSE on the action), a softmax actor (``continuous=False``:
    weighted log-likelihood) or a GaussianActorNetwork (``continuous="gaussian"``: weighted
    ``get_log_probability`` of the dataset action, :231-236, actor_networks.py:593-629)."""

    def __init__(self, actor_sd, value_sd, critic_sd, critic_target_sd, continuous: bool,
                 low: Tensor = None, high: Tensor = None, gamma: float = 0.99, tau: float = 0.05,
                 lr: float = 1e-3, expectile: float = 0.5, temperature: float = 0.5,
                 adv_clamp: float = 100.0) -> None:
        self.actor = _layers(actor_sd)
        self.head = []
        if continuous == "gaussian":      # trunk (last activation relu) + fc_mu / fc_std
            self.head = [actor_sd[k].clone().requires_grad_(True) for k in
                         ("fc_mu.weight", "fc_mu.bias", "fc_std.weight", "fc_std.bias")]
        self.value = _layers(value_sd)
        self.c = [_layers(critic_sd, f"_critic_{i}._model.") for i in (1, 2)]
        self.ct = [[(w.detach().clone(), b.detach().clone()) for w, b in
                    _layers(critic_target_sd, f"_critic_{i}._model.")] for i in (1, 2)]
        self.continuous, self.low, self.high = continuous, low, high
        self.gamma, self.tau = gamma, tau
        self.expectile, self.temperature, self.adv_clamp = expectile, temperature, adv_clamp
        self.opt_v = _adamw(_flat(self.value), lr)
        self.opt_a = _adamw(_flat(self.actor) + self.head, lr)
        self.opt_c = _adamw(_flat(self.c[0]) + _flat(self.c[1]), lr)

    @staticmethod
    def q(layers, state, action) -> Tensor:
        return mlp(layers, torch.cat([state, action], dim=-1)).view(-1)

    def gaussian_log_prob(self, state: Tensor, action: Tensor) -> Tensor:
        """GaussianActorNetwork.get_log_probability (actor_networks.py:593-629), (B,)."""
        x = mlp(self.actor, state, last_relu=True)
        mean = torch.nn.functional.linear(x, self.head[0], self.head[1])
        log_std = torch.tanh(torch.nn.functional.linear(x, self.head[2], self.head[3]))
        log_std = -5 + 0.5 * (2 - (-5)) * (log_std + 1)
        std = log_std.exp()
        n = torch.clip((((action - self.low) / (self.high - self.low)) * 2) - 1, -1 + 1e-6, 1 - 1e-6)
        u = torch.atanh(n)
        log_prob = -((u - mean) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))
        log_prob = log_prob - torch.log(((self.high - self.low) / 2) * (1 - n.pow(2)) + 1e-6)
        return log_prob.sum(dim=1)

    def learn_batch(self, batch: Dict[str, Tensor]) -> Dict[str, float]:
        """Draws the two target-critic indices from torch's global generator like the reference."""
        s, a, r, term, ns = (batch[k] for k in ("state", "action", "reward", "terminated", "next_state"))
        with torch.no_grad():
            tq = [self.q(self.ct[0], s, a), self.q(self.ct[1], s, a)]
        # ---- value (:186-196, :271-285)
        tqv = tq[int(torch.randint(0, 2, (1,)).item())]
        v = mlp(self.value, s).view(-1)
        d = tqv - v
        value_loss = (torch.where(d > 0, self.expectile, 1 - self.expectile) * d.pow(2)).mean()
        # ---- critic (:248-269)
        with torch.no_grad():
            y = (mlp(self.value, ns).view(-1) * self.gamma * (1 - term.float())) + r
        mse = torch.nn.MSELoss()
        critic_loss = (mse(self.q(self.c[0], s, a), y) + mse(self.q(self.c[1], s, a), y)) / 2.0
        # ---- actor (:197-246)
        tqa = tq[int(torch.randint(0, 2, (1,)).item())]
        with torch.no_grad():
            adv = torch.clamp(torch.exp((tqa - mlp(self.value, s).view(-1)) * self.temperature),
                              max=self.adv_clamp)
        if self.continuous == "gaussian":
            actor_loss = -(adv * self.gaussian_log_prob(s, a)).mean()
        elif self.continuous:
            z = mlp(self.actor, s)
            pred = (((self.high - self.low) * (torch.tanh(z) + 1.0)) / 2) + self.low
            actor_loss = (adv * (pred - a).pow(2).mean(dim=1)).mean()
        else:
            z = mlp(self.actor, s)
            p = torch.softmax(z, dim=-1)
            idx = torch.argmax(a, dim=1).unsqueeze(-1)
            actor_loss = -(adv * torch.log(torch.gather(p, 1, idx).view(-1))).mean()
        for o in (self.opt_v, self.opt_a, self.opt_c):
            o.zero_grad()
        (value_loss + critic_loss + actor_loss).backward()
        self.opt_v.step()
        self.opt_a.step()
        self.opt_c.step()
        with torch.no_grad():
            for net, tgt in zip(self.c, self.ct):
                for (w, b), (tw, tb) in zip(net, tgt):
                    tw.copy_(self.tau * w + (1.0 - self.tau) * tw)
                    tb.copy_(self.tau * b + (1.0 - self.tau) * tb)
        return {"value_loss": value_loss.item(), "actor_loss": actor_loss.item(),
                "critic_loss": critic_loss.item()}


# --------------------------------------------------------------------------------------
# DDPG / TD3
# --------------------------------------------------------------------------------------
class DdpgOracle:
    """DeepDeterministicPolicyGradient (ddpg.py:106-156) and, with ``td3=True``, TD3
    (td3.py:106-201): tanh actor + target, twin critics + targets, torch autograd + AdamW(amsgrad)."""

    def __init__(self, actor_sd, actor_target_sd, critic_sd, critic_target_sd, low: Tensor,
                 high: Tensor, gamma: float = 0.99, tau: float = 0.005, lr: float = 1e-3,
                 td3: bool = False, actor_update_freq: int = 2, noise_clip: float = 0.5) -> None:
        self.actor = _layers(actor_sd)
        self.actor_t = [(w.detach().clone(), b.detach().clone()) for w, b in _layers(actor_target_sd)]
        self.c = [_layers(critic_sd, f"_critic_{i}._model.") for i in (1, 2)]
        self.ct = [[(w.detach().clone(), b.detach().clone()) for w, b in
                    _layers(critic_target_sd, f"_critic_{i}._model.")] for i in (1, 2)]
        self.low, self.high = low, high
        self.gamma, self.tau = gamma, tau
        self.td3, self.freq, self.clip = td3, actor_update_freq, noise_clip
        self.opt_a = _adamw(_flat(self.actor), lr)
        self.opt_c = _adamw(_flat(self.c[0]) + _flat(self.c[1]), lr)
        self.training_steps = 0
        self.last_actor_loss = 0.0

    def policy(self, layers, state: Tensor) -> Tensor:
        """VanillaContinuousActorNetwork.sample_action (actor_networks.py:448-485)."""
        n = torch.tanh(mlp(layers, state))
        return (((self.high - self.low) * (n + 1.0)) / 2) + self.low

    @staticmethod
    def q(layers, state, action) -> Tensor:
        return mlp(layers, torch.cat([state, action], dim=-1)).view(-1)

    def _soft(self, net, tgt, tau=None) -> None:
        with torch.no_grad():
            for (w, b), (tw, tb) in zip(net, tgt):
                tw.copy_(self.tau * w + (1.0 - self.tau) * tw)
                tb.copy_(self.tau * b + (1.0 - self.tau) * tb)

    def actor_step(self, s: Tensor) -> float:
        loss = -self.q(self.c[0], s, self.policy(self.actor, s)).mean()     # ddpg.py:106-121
        self.opt_a.zero_grad()
        loss.backward()
        self.opt_a.step()
        return loss.item()

    def critic_step(self, batch: Dict[str, Tensor], noise) -> float:
        s, a, r, term, ns = (batch[k] for k in ("state", "action", "reward", "terminated", "next_state"))
        self.opt_c.zero_grad()
        with torch.no_grad():
            na = self.policy(self.actor_t, ns)
            if noise is not None:                                            # td3.py:151-175
                n = torch.clamp(noise, -self.clip, self.clip)
                n = n * (self.high - self.low) / 2
                na = torch.clamp(na + n, self.low, self.high)
            nq = torch.minimum(self.q(self.ct[0], ns, na), self.q(self.ct[1], ns, na))
            y = (nq * self.gamma * (1 - term.float())) + r
        mse = torch.nn.MSELoss()
        loss = (mse(self.q(self.c[0], s, a), y) + mse(self.q(self.c[1], s, a), y)) / 2.0
        loss.backward()
        self.opt_c.step()
        return loss.item()

    def learn_batch(self, batch: Dict[str, Tensor], noise=None) -> Dict[str, float]:
        due = (not self.td3) or self.training_steps % self.freq == 0
        if due:
            self.last_actor_loss = self.actor_step(batch["state"])
        report = {"actor_loss": self.last_actor_loss,
                  "critic_loss": self.critic_step(batch, noise if self.td3 else None)}
        if due:
            for net, tgt in zip(self.c, self.ct):
                self._soft(net, tgt)
            self._soft(self.actor, self.actor_t)
        return report


# --------------------------------------------------------------------------------------
# neural-linear contextual bandit
# --------------------------------------------------------------------------------------
class NeuralLinearOracle:
    """NeuralLinearBandit.learn_batch (policy_learners/contextual_bandits/neural_linear_bandit.py
    :159-225) over NeuralLinearRegression (neural_networks/contextual_bandit/
    neural_linear_regression.py:125-157) and LinearRegression.learn_batch / calculate_coefs /
    calculate_sigma (linear_regression.py:192-219, :252-270)."""

    def __init__(self, model_sd, lr: float = 3e-4, l2_reg_lambda: float = 1.0,
                 loss_type: str = "mse", output_activation: str = "linear",
                 hidden_activation: str = "relu", nn_e2e: bool = True, force_pinv: bool = False,
                 use_batch_norm: bool = False, dropout_ratio: float = 0.0,
                 use_skip_connections: bool = False) -> None:
        # force_pinv (linear_regression.py:138-157): torch.linalg.pinv(hermitian=True) instead of inv
        self.force_pinv = force_pinv
        # nn_e2e=False (neural_linear_regression.py:100-105, :140-147): mu from the LinUCB regression's
        # coefficients (buffers: the loss reaches the trunk through them, linear_layer_e2e gets no
        # gradient and AdamW skips it)
        self.nn_e2e = nn_e2e
        # mlp_block's remaining options (common/utils.py:113-131, :142-150): nn.Dropout between
        # (LayerNorm and) activation, nn.BatchNorm1d AFTER the activation (training mode: the reference
        # never switches to eval), ResidualWrapper around a layer whose in / out widths agree (its
        # state-dict keys then carry a `module.` level)
        self.p_drop, self.use_bn = float(dropout_ratio), bool(use_batch_norm)
        self.masks: List[Tensor] = []      # dropout keep masks of the NEXT forward, hidden layer order
        if use_batch_norm or dropout_ratio > 0 or use_skip_connections:
            self._init_general(model_sd)
        else:
            self.wrapped, self.bn = None, None
        self.trunk = self.trunk if getattr(self, "wrapped", None) else _layers(model_sd, "_nn_layers._model.")
        # mlp_block's other forms (common/utils.py:75-152): nn.LayerNorm between a hidden Linear and
        # its activation when the state dict has `{i}.1.weight`; the hidden activation by name
        pre = "_nn_layers._model."
        bases = self._bases if getattr(self, "wrapped", None) else [f"{pre}{i}." for i in range(len(self.trunk))]
        self.norms = [(model_sd[f"{b}1.weight"].clone().requires_grad_(True),
                       model_sd[f"{b}1.bias"].clone().requires_grad_(True))
                      if (f"{b}1.weight" in model_sd and f"{b}1.running_mean" not in model_sd) else None
                      for b in bases[:-1]]
        from oracle.pearl_oracle import _HIDDEN_ACTS
        self.hidden_act = _HIDDEN_ACTS[hidden_activation]
        self.e2e = model_sd["linear_layer_e2e.weight"].clone().requires_grad_(True)
        d = self.e2e.shape[1]
        self.A, self.b = torch.zeros(d + 1, d + 1), torch.zeros(d + 1)
        self.sum_weight = torch.zeros(1)
        self.inv_A, self.coefs = torch.zeros(d + 1, d + 1), torch.zeros(d + 1)
        self.lam = l2_reg_lambda
        bn_params = [t for b in (self.bn or []) if b for t in (b["weight"], b["bias"])]
        self.opt = torch.optim.AdamW(_flat(self.trunk) + [t for n in self.norms if n for t in n] + bn_params
                                     + [self.e2e], lr=lr, amsgrad=True)
        # LossType.function() (neural_networks/common/utils.py:60-72) and the model's output
        # activation (neural_linear_regression.py:79-81)
        self.criterion = {"mse": torch.nn.functional.mse_loss, "mae": torch.nn.functional.l1_loss,
                          "cross_entropy": torch.nn.functional.binary_cross_entropy}[loss_type]
        self.out_act = {"linear": lambda z: z, "sigmoid": torch.sigmoid}[output_activation]

    def _init_general(self, sd) -> None:
        pre, i = "_nn_layers._model.", 0
        self.trunk, self.wrapped, bases = [], [], []
        while any(k.startswith(f"{pre}{i}.") for k in sd):
            wrapped = f"{pre}{i}.module.0.weight" in sd
            base = f"{pre}{i}." + ("module." if wrapped else "")
            self.trunk.append((sd[base + "0.weight"].clone().requires_grad_(True),
                               sd[base + "0.bias"].clone().requires_grad_(True)))
            self.wrapped.append(wrapped)
            bases.append(base)
            i += 1
        self.wrapped = self.wrapped or [False]
        self._bases = bases
        self.bn = []
        for li, base in enumerate(bases[:-1]):
            has_ln = (base + "1.weight") in sd and sd[base + "1.weight"].ndim == 1 and \
                (base + "1.running_mean") not in sd
            idx = 1 + int(has_ln) + int(self.p_drop > 0) + 1
            if self.use_bn:
                b = f"{base}{idx}."
                self.bn.append(dict(weight=sd[b + "weight"].clone().requires_grad_(True),
                                    bias=sd[b + "bias"].clone().requires_grad_(True),
                                    running_mean=sd[b + "running_mean"].clone(),
                                    running_var=sd[b + "running_var"].clone(),
                                    nbt=sd[b + "num_batches_tracked"].clone(), key=b))
            else:
                self.bn.append(None)

    def features(self, x: Tensor, train: bool = True) -> Tensor:
        if getattr(self, "wrapped", None):
            return self._features_general(x, train)
        # the trunk's own last layer has no activation (and no LayerNorm)
        for i, (w, b) in enumerate(self.trunk):
            x = torch.nn.functional.linear(x, w, b)
            if i + 1 < len(self.trunk):
                if self.norms[i] is not None:
                    mu = x.mean(dim=-1, keepdim=True)
                    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
                    x = (x - mu) / torch.sqrt(var + 1e-5) * self.norms[i][0] + self.norms[i][1]
                x = self.hidden_act(x)
        return x

    def _features_general(self, x: Tensor, train: bool) -> Tensor:
        F = torch.nn.functional
        for i, (w, b) in enumerate(self.trunk):
            inp = x
            x = F.linear(x, w, b)
            if i + 1 < len(self.trunk):
                if self.norms[i] is not None:
                    mu = x.mean(dim=-1, keepdim=True)
                    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
                    x = (x - mu) / torch.sqrt(var + 1e-5) * self.norms[i][0] + self.norms[i][1]
                if self.p_drop > 0 and train:
                    keep = self.masks.pop(0)               # the reference's draw (torch's dropout:
                    x = x * keep.div(1.0 - self.p_drop)    #  input * bernoulli(1 - p).div_(1 - p))
                x = self.hidden_act(x)
                if self.bn[i] is not None:
                    bn = self.bn[i]
                    x = F.batch_norm(x, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"],
                                     training=train, momentum=0.1, eps=1e-5)
                    if train:
                        bn["nbt"] += 1
            if self.wrapped[i]:
                x = inp + x
        return x

    def learn_batch(self, x: Tensor, y: Tensor, w) -> Dict[str, Tensor]:
        f = self.features(x)
        if self.nn_e2e:
            pred = self.out_act(torch.nn.functional.linear(f, self.e2e))
        else:   # LinearRegression.forward (linear_regression.py:221-250): [1 | f] coefs
            pred = self.out_act(torch.matmul(torch.cat((torch.ones(x.shape[0], 1), f), dim=-1),
                                             self.coefs.detach()).unsqueeze(-1))
        weight = torch.ones_like(y) if w is None else w
        loss = self.criterion(pred.view(y.shape), y, reduction="none")
        loss = (loss * weight).sum() / weight.sum()
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        X = torch.cat((torch.ones(x.shape[0], 1), f.detach()), dim=-1)
        wc, yc = weight.unsqueeze(-1), y.unsqueeze(-1)
        dA = torch.matmul(X.t(), X * wc)
        self.A += (dA + dA.t()) / 2
        self.b += torch.matmul(X.t(), yc * wc).squeeze(-1)
        self.sum_weight += weight.sum()
        M = self.A + self.lam * torch.eye(self.A.shape[0])
        self.inv_A = torch.linalg.pinv(M, hermitian=True) if self.force_pinv else torch.linalg.inv(M)
        self.coefs = torch.matmul(self.inv_A, self.b)
        return {"loss": loss.detach(), "prediction": pred.detach()}

    @torch.no_grad()
    def sigma(self, x: Tensor, train: bool = True) -> Tensor:
        X = torch.cat((torch.ones(x.shape[0], 1), self.features(x, train)), dim=-1)
        return torch.sqrt((torch.matmul(X, self.inv_A) * X).sum(-1))


# --------------------------------------------------------------------------------------
# SquareCB
# --------------------------------------------------------------------------------------
def squarecb_probs(values: Tensor, gamma: float, clamp_values: bool = False, reward_lb: float = 0.0,
                   reward_ub: float = 1.0) -> Tensor:
    """SquareCBExploration.act's probability table for ONE context (squarecb_exploration.py:71-92;
    the batch size its whole-matrix complementary sum, :90, is a distribution for): values (1, A)."""
    A = values.shape[-1]
    values = values.view(-1, A)
    assert values.shape[0] == 1
    if clamp_values:
        values = torch.clamp(values, min=reward_lb, max=reward_ub)
    max_val, max_indices = torch.max(values, dim=1)
    prob = torch.div(1.0, A + gamma * (max_val - values))
    prob[0, max_indices[0]] = 0.0
    prob[0, max_indices[0]] = 1.0 - torch.sum(prob)
    return prob
