"""CLI of the multi-t reconstruction path.

Flag names, types and defaults are the reference's (/root/reference/reconstruct.py:7-141) so
existing launch scripts keep working; the flags the reference parses but never reads
(--eval_checkpoint, --augmentation, --cache_data, --num_workers, --num_inference_steps;
SURVEY Q1/Q12) are accepted and ignored in the same way.  Extensions default to the
reference's behaviour.

    python reconstruct.py --output_dir=... --model_name=fashionmnist --validation_ids=... \
        --in_ids=... --out_ids=a.csv,b_vflip.csv --is_grayscale=1 \
        --beta_schedule=scaled_linear_beta --beta_start=0.0015 --beta_end=0.0195 \
        --inference_skip_factor=4
    torchrun --nproc_per_node=8 --master-addr 127.0.0.1 reconstruct.py ...   # one rank per MI355X
"""

import argparse
import ast

# (flag, type, default, help)
_FLAGS = [
    ("seed", int, 2, "seed of the per-image noise inputs"),
    ("output_dir", str, None, "root directory holding <model_name>/checkpoint.pth"),
    ("model_name", str, None, "run directory name"),
    ("validation_ids", str, None, "id file (or synthetic: spec) of the validation set"),
    ("in_ids", str, None, "id file of the in-distribution test set"),
    ("out_ids", str, None, "comma list of OOD id files; a _vflip / _hflip suffix adds the flip"),
    ("spatial_dimension", int, 2, "2 or 3"),
    ("image_size", None, None, "resize to this extent"),
    ("image_roi", ast.literal_eval, None, "central crop, tuple, -1 keeps a dimension"),
    ("latent_pad", ast.literal_eval, None, "F.pad-style padding of the latent"),
    ("vqvae_checkpoint", None, None, "VQ-VAE checkpoint for latent diffusion"),
    ("ddpm_checkpoint_epoch", None, None, "use checkpoint_<epoch>.pth instead of checkpoint.pth"),
    ("prediction_type", None, "epsilon", "epsilon or v_prediction"),
    ("model_type", None, "small", "small or big"),
    ("beta_schedule", None, "linear", "linear[_beta] | scaled_linear[_beta] | sigmoid[_beta] | cosine"),
    ("beta_start", float, 1e-4, "first beta"),
    ("beta_end", float, 2e-2, "last beta"),
    ("b_scale", float, 1, "data scale applied before noising"),
    ("snr_shift", float, 1, "SNR shift factor of the schedule"),
    ("simplex_noise", int, 0, "not on this path (must stay 0)"),
    ("batch_size", int, 256, "images per batch"),
    ("augmentation", int, 0, "ignored (as in the reference)"),
    ("cache_data", int, 1, "ignored: data is always cached"),
    ("num_workers", int, 8, "ignored: ingest is in-process"),
    ("first_n_val", None, None, "truncate the validation set"),
    ("first_n", None, None, "truncate every other set"),
    ("eval_checkpoint", None, None, "ignored (as in the reference)"),
    ("drop_last", None, False, "drop a ragged last batch"),
    ("is_grayscale", int, 0, "1-channel data"),
    ("run_val", int, 1, "score the validation set"),
    ("run_in", int, 1, "score the in-distribution set"),
    ("run_out", int, 1, "score the OOD sets"),
    ("num_inference_steps", int, 100, "ignored unless --honour_num_inference_steps=1 (reference hard-codes 100)"),
    ("inference_skip_factor", int, 1, "use every k-th timestep as a reconstruction start"),
]

_EXTENSIONS = [
    ("honour_num_inference_steps", int, 0, "1: really use --num_inference_steps (Q1)"),
    ("reset_scheduler_per_t", int, 0, "1: clear the PLMS history before each t-start (deviation, Q3)"),
    ("use_proj_attn", int, 0, "1: apply AttentionBlock.proj_attn (SURVEY A.3 open point)"),
    ("lpips_weights", None, None, "state_dict file with LPIPS-AlexNet weights"),
]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name, typ, default, text in _FLAGS:
        kw = {"default": default, "help": text}
        if typ not in (None, str):
            kw["type"] = typ
        parser.add_argument(f"--{name}", **kw)
    ext = parser.add_argument_group("extensions (defaults reproduce the reference)")
    for name, typ, default, text in _EXTENSIONS:
        kw = {"default": default, "help": text}
        if typ is not None:
            kw["type"] = typ
        ext.add_argument(f"--{name}", **kw)
    ext.add_argument("--timestep_list", default="monai", choices=["monai", "diffusers"],
                     help="100-entry or 101-entry PLMS timestep list (Q9)")
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)


if __name__ == "__main__":
    args = parse_args()
    from ddpm_ood_amd.trainer import Reconstruct

    recon = Reconstruct(args)
    recon.reconstruct(args)
    import torch.distributed as dist

    if dist.is_initialized():
        dist.destroy_process_group()
