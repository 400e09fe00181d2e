"""Where a native-loop learn() call spends its wall time (debug aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
sys.argv = [sys.argv[0]]
from host_bound import make_sac, make_td3
from pearl_amd import _native as N
from pearl_amd.replay_buffers.basic_replay_buffer import TensorBasedReplayBuffer

log = []
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); log.append((name, (time.perf_counter() - t0) * 1e3)); return r
    return w
def install():
    torch.randn = timed("randn", torch.randn)
    torch.zeros = timed("zeros", torch.zeros)
    torch.Tensor.tolist = timed("tolist", torch.Tensor.tolist)
    TensorBasedReplayBuffer.presample = timed("presample", TensorBasedReplayBuffer.presample)
    from pearl_amd import ContinuousSoftActorCritic as S
    from pearl_amd.policy_learners.sequential_decision_making.flat_mlp import FlatMlp
    S._nets = timed("_nets", S._nets)
    S._arena_loop_plan = timed("plan", S._arena_loop_plan)
    S._step_args = timed("step_args", S._step_args)
    S._learn_native_loop = timed("native_loop", S._learn_native_loop)
    FlatMlp.stepped_natively = timed("stepped", FlatMlp.stepped_natively)
    FlatMlp.leave_learn_loop = staticmethod(timed("leave", FlatMlp.leave_learn_loop))
for name, mk in (("sac", make_sac),):
    learn = mk(300)
    lib = N.lib()
    real = lib.pa_sac_learn
    for call in range(9):
        if call == 1:
            install()
            if os.environ.get("GCOFF") == "1":
                import gc
                gc.collect(); gc.disable()
            lib.pa_sac_learn = timed("pa_sac_learn", real)
        log.clear()
        if os.environ.get("PRESYNC") == "1" and call >= 1:
            orig = lib.pa_sac_learn
            def w2(*a, _o=orig):
                rc = _o(*a); t = time.perf_counter(); torch.cuda.current_stream().synchronize(); log.append(("streamsync", (time.perf_counter() - t) * 1e3)); return rc
            lib.pa_sac_learn = w2
        t0 = time.perf_counter(); learn(); t1 = time.perf_counter(); torch.cuda.synchronize()
        print(name, "call", call, "learn() ms", round((t1 - t0) * 1e3, 2), [(n, round(t, 2)) for n, t in log if t > 0.5])
        if os.environ.get("PRESYNC") == "1" and call >= 1:
            lib.pa_sac_learn = orig
