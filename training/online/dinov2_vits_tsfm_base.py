#!/usr/bin/env python3
"""Drop-in for the reference's training entry point at the SAME path and with the same command line
(/root/reference/training/online/dinov2_vits_tsfm_base.py:395-402 ``fire.Fire(DinoV2ViTSTSFMBaseRunner)``;
/root/reference/scripts/train.sh:116-136, README.md:239-257):

    python training/online/dinov2_vits_tsfm_base.py train --il_ckpt_path IL.ckpt --num_train_processes 32 \\
        --output_dir out --dataset_dir data/fifteen/PickupType --cost_limit 2.31964 --tag PickupType [--checkpoint X.pt ...]

``fire`` semantics kept: the command (``train``) may come before or after the flags, flags are the dataclass fields of
``DinoV2ViTSTSFMBaseParams`` / ``BaseConfigParams`` / ``OnPolicyRunnerMixin`` given as ``--flag value`` or ``--flag=value``.  The run
itself is the MI355X engine (safevla_amd.train): synthetic environments in place of AI2-THOR (simulator off the critical path), the
task type inferred from ``--tag`` / ``--dataset_dir`` like the reference infers its task sampler from the dataset directory.
Multi-GPU: ``torchrun --nproc-per-node N training/online/dinov2_vits_tsfm_base.py train ...`` (one rank per GPU, RCCL).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cmds = [a for a in argv if a in ("train", "test")]
    if not cmds:
        raise SystemExit("usage: dinov2_vits_tsfm_base.py train [--flag value ...]   (fire-style; see the module docstring)")
    if cmds[0] == "test":
        raise SystemExit("`test` drives AllenAct's simulator-bound evaluation runner; use safevla_amd.agent.InferenceAgentVIDA for evaluation")
    rest = [a for a in argv if a not in ("train", "test")]
    from safevla_amd import train

    return train.main(["train"] + rest)


if __name__ == "__main__":
    raise SystemExit(main())
