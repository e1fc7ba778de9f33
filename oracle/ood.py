"""Oracle restatement of the scoring stage, /root/reference/ood_detection.py:40-223.

TEST INFRASTRUCTURE (see oracle/__init__.py).  pandas / scikit-learn exactly as the
reference uses them (both ARE installed here, so this stage is pinned to the real
libraries): drop_duplicates(filename,t) keep-first (:54,144-145), strict MIN_T < t < MAX_T
(:59-61), per-t val mean / pandas std ddof=1 (:150-161), mean over t per (filename,type)
(:174), roc_auc_score(in=0, out=1) on z_score_mse (:195-206).
"""

from __future__ import annotations

import pandas as pd
from sklearn.metrics import roc_auc_score


def z_scores_and_auroc(df_val: pd.DataFrame, df_in: pd.DataFrame, df_out: pd.DataFrame,
                       max_t: int = 1000, min_t: int = 0, plot_target: str = "mse"):
    df_val = df_val.drop_duplicates(subset=["filename", "t"], keep="first")
    t_values = df_val["t"].unique()
    t_values = t_values[t_values < max_t]
    t_values = t_values[min_t < t_values]
    df_val = df_val[df_val["t"].isin(t_values)]
    t_values = df_val["t"].unique()
    df_in = df_in.drop_duplicates(subset=["filename", "t"], keep="first")
    df_out = df_out.drop_duplicates(subset=["filename", "t"], keep="first")
    df_in = df_in[df_in["t"].isin(t_values)]
    df_out = df_out[df_out["t"].isin(t_values)]
    df = pd.concat((df_in, df_out))
    for target in ["perceptual_difference", "mse"]:
        agg = (df_val.groupby(["t"]).agg({target: ["mean", "std"]})[target].reset_index()
               .rename({"mean": f"val_mean_{target}", "std": f"val_std_{target}"}, axis=1))
        df = df.merge(agg, on=["t"], how="left")
        df[f"z_score_{target}"] = (df[target] - df[f"val_mean_{target}"]) / df[f"val_std_{target}"]
    if plot_target == "mse+perceptual":
        df["z_score_mse+perceptual"] = df["z_score_mse"] + df["z_score_perceptual_difference"]
    target = f"z_score_{plot_target}"
    df_mean = df.groupby(["filename", "type"]).mean().reset_index()
    s_in = df_mean.loc[df_mean["type"] == "in"][[target]].values.tolist()
    s_out = df_mean.loc[df_mean["type"] == "out"][[target]].values.tolist()
    auc = roc_auc_score([0] * len(s_in) + [1] * len(s_out), s_in + s_out)
    return df, df_mean, auc
