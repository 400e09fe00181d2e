"""GPU debug: where do the fused pa_dqn_learn loop and the generic per-step loop diverge?"""
import os, sys, random
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_dqn import make_learner
from test_gpu_replay import fill_arena_buffer
from pearl_amd.policy_learners.policy_learner import PolicyLearner

for name in ["tiny_dynamic", "cfg1_cartpole_shape", "cfg2_shape_small_batch"]:
    fx = torch.load(os.path.join(REPO, "tests", "golden", f"dqn_{name}.pt"), map_location="cpu", weights_only=False)
    for rounds in [1, 2, 3, 8, 9, 10, 12]:
        a, b = make_learner(fx, training_rounds=rounds), make_learner(fx, training_rounds=rounds)
        rb = fill_arena_buffer(fx, "python")
        random.seed(4); ra = a.learn(rb)
        random.seed(4); rbr = PolicyLearner.learn(b, rb)
        torch.cuda.synchronize()
        out = []
        for (k, pa), (_, pb) in zip(a._Q.state_dict().items(), b._Q.state_dict().items()):
            out.append(f"{k.split('.')[1]}{k[-1]}:{(pa-pb).abs().max().item():.2e}")
        for (k, pa), (_, pb) in zip(a._Q_target.state_dict().items(), b._Q_target.state_dict().items()):
            out.append(f"T{k.split('.')[1]}{k[-1]}:{(pa-pb).abs().max().item():.2e}")
        g = [(pa.grad - pb.grad).abs().max().item() for pa, pb in zip(a._Q.parameters(), b._Q.parameters())]
        print(name, "rounds", rounds, "loss_eq", ra["loss"] == rbr["loss"], " ".join(out), "grad", ["%.1e" % x for x in g], flush=True)
