#!/usr/bin/env python3
"""Where do the HIP gradients sit between the reference's fp32 (MKL) and float64?  First learn_batch of
the bandit cfg5 long-run fixture: per-layer gradient error of (a) torch fp32 autograd on the CPU and
(b) the HIP step, both against float64 autograd, relative to max |g| and as rms / rms(g)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import fixture_inputs as FI
from pearl_amd import NeuralLinearBandit, TransitionBatch
from pearl_amd import _native as N

fx = torch.load("tests/golden/bandit_cfg5_steps20.pt", map_location="cpu", weights_only=False)
cfg = fx["config"]
F, B = cfg["F"], cfg["B"]
x = FI.bandit_contexts(cfg, 0)
r = fx["batches"][0]["reward"]
sd = fx["model0"]

def grads(dtype):
    ws = [(sd[f"_nn_layers._model.{i}.0.weight"].to(dtype).clone().requires_grad_(True),
           sd[f"_nn_layers._model.{i}.0.bias"].to(dtype).clone().requires_grad_(True)) for i in range(3)]
    e2e = sd["linear_layer_e2e.weight"].to(dtype).clone().requires_grad_(True)
    h = x.to(dtype)
    for i, (w, b) in enumerate(ws):
        h = torch.nn.functional.linear(h, w, b)
        if i < 2:
            h = torch.relu(h)
    pred = torch.nn.functional.linear(h, e2e).view(-1)
    loss = ((pred - r.to(dtype)) ** 2).mean()
    loss.backward()
    return [t.grad for wb in ws for t in wb] + [e2e.grad]

g64, g32 = grads(torch.float64), grads(torch.float32)
for split in (-1, 0):
    N.check(N.lib().pa_debug_set_dw_split(split))
    pl = NeuralLinearBandit(feature_dim=F, hidden_dims=cfg["hidden"], batch_size=B, learning_rate=1e-3)
    pl.model.load_state_dict(sd)
    pl.to("cuda:0")
    pl.learn_batch(TransitionBatch(state=x.cuda(), action=torch.zeros(B, 1, device="cuda:0"), reward=r.cuda()))
    torch.cuda.synchronize()
    ghip = [p.grad.detach().cpu() for p in list(pl.model._nn_layers.parameters()) + [pl.model.linear_layer_e2e.weight]]
    print(f"dw_split mode {split}:  layer | max|g| | torch fp32 err/max|g|, rms/rms | HIP err/max|g|, rms/rms")
    for i, (a64, a32, ah) in enumerate(zip(g64, g32, ghip)):
        m, rms = float(a64.abs().max()), float(a64.pow(2).mean().sqrt())
        e32, eh = (a32.double() - a64), (ah.double() - a64)
        print(f"  {i}: {m:.3e} | {float(e32.abs().max())/m:.2e} {float(e32.pow(2).mean().sqrt())/rms:.2e} | "
              f"{float(eh.abs().max())/m:.2e} {float(eh.pow(2).mean().sqrt())/rms:.2e}   "
              f"smallest |g64| quantiles 1%: {float(a64.abs().flatten().kthvalue(max(1, a64.numel()//100)).values):.2e}")
N.check(N.lib().pa_debug_set_dw_split(-1))
