"""CLI of the DDPM training loop: flag names, types and defaults of
/root/reference/train_ddpm.py:7-83, so the README's training commands keep working.

    python train_ddpm.py --output_dir=... --model_name=fashionmnist --training_ids=... --validation_ids=... \
        --is_grayscale=1 --n_epochs=300 --beta_schedule=scaled_linear_beta --beta_start=0.0015 --beta_end=0.0195
    torchrun --nproc_per_node=8 --master-addr 127.0.0.1 train_ddpm.py ...   # one rank per MI355X
"""

import argparse
import ast

_FLAGS = [
    ("seed", int, 2), ("output_dir", str, None), ("model_name", str, None), ("training_ids", str, None),
    ("validation_ids", str, None), ("spatial_dimension", int, 2), ("image_size", None, None),
    ("image_roi", ast.literal_eval, None), ("latent_pad", ast.literal_eval, None), ("vqvae_checkpoint", None, None),
    ("prediction_type", None, "epsilon"), ("model_type", None, "small"), ("beta_schedule", None, "linear_beta"),
    ("beta_start", float, 1e-4), ("beta_end", float, 2e-2), ("b_scale", float, 1), ("snr_shift", float, 1),
    ("simplex_noise", int, 0), ("batch_size", int, 512), ("n_epochs", int, 300), ("eval_freq", int, 10),
    ("augmentation", int, 1), ("num_workers", int, 8), ("cache_data", int, 1), ("checkpoint_every", int, 100),
    ("ddpm_checkpoint_epoch", None, None), ("is_grayscale", int, 0), ("quick_test", int, 0),
]


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name, typ, default in _FLAGS:
        kw = {"default": default}
        if typ not in (None, str):
            kw["type"] = typ
        parser.add_argument(f"--{name}", **kw)
    ext = parser.add_argument_group("extensions (defaults: fp32 training)")
    ext.add_argument("--amp", type=int, default=0,
                     help="1: fp16 autocast + GradScaler as the reference trains (ddpm_trainer.py:96-109); default 0 = fp32")
    return parser.parse_args(argv)


if __name__ == "__main__":
    args = parse_args()
    from ddpm_ood_amd.train import DDPMTrainer

    DDPMTrainer(args).train(args)
