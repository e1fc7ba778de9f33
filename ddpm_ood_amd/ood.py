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


def score(results_df_val: pd.DataFrame, results_df_in: pd.DataFrame, results_df_out: pd.DataFrame,
          max_t: int = 1000, min_t: int = 0, plot_target: str = "mse"):
    results_df_val = results_df_val.drop_duplicates(subset=["filename", "t"], keep="first")
    t_values = results_df_val["t"].unique()
    t_values = t_values[(t_values < max_t)]
    t_values = t_values[(min_t < t_values)]
    results_df_val = results_df_val[results_df_val["t"].isin(t_values)]
    t_values = results_df_val["t"].unique()
    results_df_in = results_df_in.drop_duplicates(subset=["filename", "t"], keep="first")
    results_df_out = results_df_out.drop_duplicates(subset=["filename", "t"], keep="first")
    results_df_in = results_df_in[results_df_in["t"].isin(t_values)]
    results_df_out = results_df_out[results_df_out["t"].isin(t_values)]
    results_df = pd.concat((results_df_in, results_df_out))
    for target in ["perceptual_difference", "mse"]:
        agg = (results_df_val.groupby(["t"]).agg({target: ["mean", "std"]})[target].reset_index()
               .rename({"mean": f"val_mean_{target}", "std": f"val_std_{target}"}, axis=1))
        results_df = results_df.merge(agg, on=["t"], how="left")
        results_df[f"z_score_{target}"] = (results_df[target] - results_df[f"val_mean_{target}"]) / \
            results_df[f"val_std_{target}"]
    if plot_target == "mse+perceptual":
        results_df["z_score_mse+perceptual"] = results_df["z_score_mse"] + results_df["z_score_perceptual_difference"]
    target = f"z_score_{plot_target}"
    results_df_mean = results_df.groupby(["filename", "type"]).mean().reset_index()
    all_scores = results_df_mean.loc[results_df_mean["type"] == "in"][[target]].values.tolist()
    all_class = [0] * len(all_scores)
    out_scores = results_df_mean.loc[results_df_mean["type"] == "out"][[target]].values.tolist()
    all_scores.extend(out_scores)
    all_class.extend([1] * len(out_scores))
    return results_df, results_df_mean, roc_auc_score(all_class, all_scores)


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
