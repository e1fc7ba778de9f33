"""Scoring stage: results_*.csv -> per-t Z-scores -> AUROC.

Mirror of /root/reference/ood_detection.py:40-223 minus the plotting (:177-192) and the
MONAI imports it only uses for a step count (:65-71, restated with this package's
scheduler).  pandas / scikit-learn semantics are the reference's: drop_duplicates keep-first
(:54,144-145), strict MIN_T < t < MAX_T (:59-61), pandas std ddof=1 (:152-161), groupby mean
over t (:174), roc_auc_score(in=0, out=1) on the MSE Z-score (:195-206).
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas as pd
from sklearn.metrics import roc_auc_score

MEDNIST = ["AbdomenCT", "BreastMRI", "ChestCT", "CXR", "Hand", "HeadCT"]


def out_datasets_for(model: str):
    """/root/reference/ood_detection.py:91-135."""
    if "fashionmnist" in model:
        return ("MNIST", "FashionMNIST_vflip", "FashionMNIST_hflip")
    if "mnist" in model:
        return ("FashionMNIST", "MNIST_vflip", "MNIST_hflip")
    if "cifar10" in model:
        return ("SVHN", "CelebA", "CIFAR10_vflip", "CIFAR10_hflip")
    if "celeba" in model.lower():
        return ("CIFAR10", "SVHN", "CelebA_vflip", "CelebA_hflip")
    if "svhn" in model:
        return ("CIFAR10", "CelebA", "SVHN_vflip", "SVHN_hflip")
    for key, name in (("abdomenct", "AbdomenCT"), ("breastmri", "BreastMRI"), ("cxr", "CXR"),
                      ("chestct", "ChestCT"), ("hand", "Hand"), ("headct", "HeadCT")):
        if key in model:
            return tuple(d for d in MEDNIST if d != name)
    if "decathlon" in model or "Task01" in model:
        return tuple(f"Task{i:02d}" for i in range(2, 11))
    raise ValueError(f"Unknown dataset to select for run_dir {model}")


def count_model_evaluations(t_values, num_inference_steps: int = 100) -> int:
    """ood_detection.py:63-71: UNet evaluations needed for a set of start points."""
    ts = np.arange(0, num_inference_steps)[::-1] * (1000 // num_inference_steps)
    return int(sum((ts <= t).sum() for t in t_values))


TARGETS = ("perceptual_difference", "mse")


def _rows(df: pd.DataFrame, keep_t=None) -> pd.DataFrame:
    """One row per (image, t-start) -- a padded DDP shard repeats images, the first copy counts -- inside the t window."""
    rows = df[~df.duplicated(subset=["filename", "t"])]
    return rows if keep_t is None else rows[rows["t"].isin(keep_t)]


def score(val: pd.DataFrame, ind: pd.DataFrame, ood: pd.DataFrame, max_t: int = 1000, min_t: int = 0, plot_target: str = "mse"):
    """Z-scores and AUROC of one in- / out-of-distribution pair of result tables against the validation table
    (what /root/reference/ood_detection.py:141-206 computes per out-dataset).

    For every t-start inside the open window (min_t, max_t): the validation set's mean and sample standard deviation of each
    similarity column; every in / out row gets z = (x - mean_t) / std_t; an image's score is the mean of its z over the
    t-starts; AUROC with in = 0, out = 1.  Returns (rows with `val_mean_*`, `val_std_*`, `z_score_*` columns,
    per-image means, AUROC).  NaN rows (a genuine overflow, written as the reference would) are skipped by the per-t
    statistics and by an image's mean; an image with no finite row makes roc_auc_score raise, as it does for the reference."""
    val_rows = _rows(val)
    window = val_rows["t"][(val_rows["t"] > min_t) & (val_rows["t"] < max_t)].unique()
    stats = val_rows[val_rows["t"].isin(window)].groupby("t")[list(TARGETS)].agg(["mean", "std"])  # NaN-skipping, ddof = 1
    table = pd.concat((_rows(ind, window), _rows(ood, window)), ignore_index=True)
    for col in TARGETS:
        mu, sigma = table["t"].map(stats[(col, "mean")]), table["t"].map(stats[(col, "std")])
        table[f"val_mean_{col}"], table[f"val_std_{col}"] = mu.to_numpy(), sigma.to_numpy()
        table[f"z_score_{col}"] = ((table[col] - mu) / sigma).to_numpy()
    if plot_target == "mse+perceptual":
        table["z_score_mse+perceptual"] = table["z_score_mse"] + table["z_score_perceptual_difference"]
    per_image = table.groupby(["filename", "type"]).mean(numeric_only=True).reset_index()
    z = per_image[f"z_score_{plot_target}"].to_numpy()
    is_ood = (per_image["type"] == "out").to_numpy()
    known = is_ood | (per_image["type"] == "in").to_numpy()
    return table, per_image, roc_auc_score(is_ood[known].astype(int), z[known])


def main(args, out_data=None):
    model = args.model_name
    run_dir = Path(args.output_dir) / model
    print(f"Run directory: {str(run_dir)}")
    out_dir = run_dir / "ood"
    out_dir.mkdir(exist_ok=True)
    results_df_val = pd.read_csv(out_dir / "results_val.csv")
    t_all = results_df_val.drop_duplicates(subset=["filename", "t"], keep="first")["t"].unique()
    t_values = t_all[(t_all < args.max_t) & (args.min_t < t_all)]
    plot_target = getattr(args, "plot_target", "mse")
    print(f"SETTING MAX_T to {args.max_t} and T_SKIP to 1 with a total of {len(t_values)} starting points "
          f"{count_model_evaluations(t_values)} model evaluations")
    print(f"Plot target is {plot_target}")
    if out_data is None:
        out_data = out_datasets_for(model)
    scores = []
    for out_dataset in out_data:
        results_df_in = pd.read_csv(out_dir / "results_in.csv")
        results_df_out = pd.read_csv(out_dir / f"results_{out_dataset}.csv")
        df, df_mean, auc = score(results_df_val, results_df_in, results_df_out, args.max_t, args.min_t, plot_target)
        n_val = results_df_val["filename"].nunique()
        print(f"n_val={n_val} n_in={df.loc[df['type'] == 'in']['filename'].nunique()} "
              f"n_out={df.loc[df['type'] == 'out']['filename'].nunique()}")
        scores.append(auc)
    for o, s in zip(out_data, scores):
        print(f"AUC for {model} vs {o}: {s * 100:.1f}")
    print(f"Average AUC: {np.mean(scores) * 100:.1f}")
    return dict(zip(out_data, scores))
