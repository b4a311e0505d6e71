"""safevla_amd -- MI355X-native PPO-Lagrangian update path for SafeVLA's actor-critic.

Only the hot path named by BASELINE.json's ``north_star`` lives here:
  csrc/      hand-written gfx950 HIP kernels behind a C ABI (include/svla.h)
  _lib.py    ctypes loader for the C-ABI shared library (fails loudly if it is missing)
  ops.py     stream-aware Python bindings + autograd glue
  model.py   host mirror of the reference's ActorCriticModel.forward API
  losses.py  SafePPOLogGrad / PPOValue / SafePPOValue / HLGaussLoss mirrors
  storage.py rollout storage with reward+cost GAE
  engine.py  the PPO-Lagrangian update loop (single- and multi-GPU)
  parallel.py / synth_env.py / api.py / lagrange.py / checkpoint.py / preproc.py / agent.py / il.py: DP plumbing, the synthetic (steppable)
  environment, the reference's output containers, lambda, checkpoint interchange, frozen ViT preprocessors, evaluation agent, IL workload
"""
__version__ = "0.2.0"
