"""CLI of the scoring stage: results_*.csv -> per-t Z-scores -> AUROC.

Flag names / defaults follow /root/reference/ood_detection.py:15-37 (--t_skip is dead there
too, Q12).  The work is in ddpm_ood_amd/ood.py.
"""

import argparse
import warnings


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__)
    for name, typ, default in (("seed", int, 2), ("output_dir", str, None), ("model_name", str, None),
                               ("max_t", int, 1000), ("min_t", int, 0), ("t_skip", int, 1)):
        p.add_argument(f"--{name}", type=typ, default=default)
    p.add_argument("--plot_target", default="mse", choices=["mse", "perceptual_difference", "mse+perceptual"],
                   help="extension: which Z-score feeds the AUROC (the reference hard-codes mse)")
    p.add_argument("--out_data", default=None,
                   help="extension: comma list of OOD set names (default: picked from --model_name)")
    return p.parse_args(argv)


if __name__ == "__main__":
    warnings.filterwarnings("ignore")
    args = parse_args()
    from ddpm_ood_amd import ood

    for model in args.model_name.split(","):
        args.model_name = model
        ood.main(args, out_data=args.out_data.split(",") if args.out_data else None)
