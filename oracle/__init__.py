"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of SafeVLA's PPO-Lagrangian hot path.

Nothing under ``safevla_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, and only as
the checker / the timed CPU baseline -- never as the thing shipped.

Pinning status (see DESIGN.md "Oracle"):

* model (``ref_model``), ``SafePPOLogGrad`` / ``PPOLogGrad`` / ``HLGaussLoss`` (``ref_loss``),
  ``PositionalEncoder`` and the llama decoder are PINNED against outputs of the reference itself,
  imported in the build container with dependency shims (``tests/golden/make_golden.py`` ->
  ``tests/golden/*.npz``);
* the T5 encoder restatement (``ref_t5``) is PINNED against ``transformers.T5EncoderModel``;
* GAE, rollout storage, ``PPOValue`` / ``SafePPOValue``, the Lagrange multiplier and the
  distributed gradient weighting live in un-vendored third-party engines (AllenAct fork,
  omnisafe 0.5.0): **parity unpinned** -- restated from the published upstream algorithms and
  property-tested only.
"""
