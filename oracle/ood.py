"""Oracle restatement of the scoring stage, /root/reference/ood_detection.py:40-223.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Deliberately NOT the pandas / scikit-learn pipeline the reference
(and therefore the product, ddpm_ood_amd/ood.py) uses: plain Python / numpy written from the definitions, so that
"product vs oracle" compares two different texts --

  * keep-first de-duplication on (filename, t)                                   ood_detection.py:54,144-145
  * strict window MIN_T < t < MAX_T on the validation t values                   :59-61
  * per-t validation mean and SAMPLE standard deviation (ddof = 1)               :150-161
  * Z = (score - mean_t) / std_t, mean over t per (filename, type)               :150-161,174
  * AUROC(in = 0, out = 1) = P(Z_out > Z_in) + P(Z_out = Z_in) / 2               :195-206  (Mann-Whitney form)

pandas appears only as the container the rows arrive in and the Z-scores leave in (same column names and row order as
the reference's frame: in rows, then out rows).  tests/test_oracle_kat.py pins this file against a third, inline
computation and tests/test_host.py pins the product's pandas / scikit-learn pipeline against it.
"""

from __future__ import annotations

import numpy as np
import pandas as pd

TARGETS = ("perceptual_difference", "mse")


def _records(df, cols):
    """Rows of a frame as tuples of Python scalars, first occurrence of every (filename, t) only."""
    seen, rows = set(), []
    arrays = [df[c].tolist() for c in cols]
    for rec in zip(*arrays):
        key = (rec[0], rec[2])  # cols = filename, type, t, ...
        if key not in seen:
            seen.add(key)
            rows.append(rec)
    return rows


def _nanmean(v) -> float:
    """groupby(...).mean() of ood_detection.py:174: NaN rows are skipped, an all-NaN group is NaN."""
    x = np.asarray(v, dtype=np.float64)
    x = x[~np.isnan(x)]
    return float(x.sum() / len(x)) if len(x) else float("nan")


def _auroc(neg, pos) -> float:
    """P(pos > neg) + P(pos == neg) / 2 by ranks (average rank on ties): the area under the ROC curve."""
    neg, pos = np.asarray(neg, dtype=np.float64), np.asarray(pos, dtype=np.float64)
    both = np.concatenate([neg, pos])
    if not np.isfinite(both).all():  # sklearn's roc_auc_score (ood_detection.py:206) refuses NaN / inf scores
        raise ValueError("Input contains NaN or infinity.")
    order = np.argsort(both, kind="mergesort")
    ranks = np.empty(len(both), dtype=np.float64)
    srt = both[order]
    i = 0
    while i < len(srt):
        j = i
        while j + 1 < len(srt) and srt[j + 1] == srt[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    u = ranks[len(neg):].sum() - len(pos) * (len(pos) + 1) / 2.0
    return float(u / (len(pos) * len(neg)))


def z_scores_and_auroc(df_val: pd.DataFrame, df_in: pd.DataFrame, df_out: pd.DataFrame,
                       max_t: int = 1000, min_t: int = 0, plot_target: str = "mse"):
    cols = ["filename", "type", "t", *TARGETS]
    val = _records(df_val, cols)
    t_keep = []
    for r in val:  # order of first appearance, like Series.unique()
        if min_t < r[2] < max_t and r[2] not in t_keep:
            t_keep.append(r[2])
    stats = {}
    for t in t_keep:
        for k, target in enumerate(TARGETS):
            x = np.array([r[3 + k] for r in val if r[2] == t], dtype=np.float64)
            x = x[~np.isnan(x)]  # pandas' groupby aggregations skip NaN (ood_detection.py:152-157)
            mean = x.sum() / len(x) if len(x) else float("nan")
            var = ((x - mean) ** 2).sum() / (len(x) - 1) if len(x) > 1 else float("nan")
            stats[(t, target)] = (mean, np.sqrt(var))
    rows = [r for r in _records(df_in, cols) if r[2] in t_keep] + [r for r in _records(df_out, cols) if r[2] in t_keep]
    out = {c: [r[i] for r in rows] for i, c in enumerate(cols)}
    for k, target in enumerate(TARGETS):
        out[f"val_mean_{target}"] = [stats[(r[2], target)][0] for r in rows]
        out[f"val_std_{target}"] = [stats[(r[2], target)][1] for r in rows]
        out[f"z_score_{target}"] = [(r[3 + k] - stats[(r[2], target)][0]) / stats[(r[2], target)][1] for r in rows]
    if plot_target == "mse+perceptual":
        out["z_score_mse+perceptual"] = [a + b for a, b in zip(out["z_score_mse"], out["z_score_perceptual_difference"])]
    df = pd.DataFrame(out)
    # mean over t per (filename, type)
    target = f"z_score_{plot_target}"
    per_image = {}
    for name, typ, z in zip(out["filename"], out["type"], out[target]):
        per_image.setdefault((name, typ), []).append(z)
    keys = sorted(per_image)
    df_mean = pd.DataFrame({"filename": [k[0] for k in keys], "type": [k[1] for k in keys],
                            target: [_nanmean(per_image[k]) for k in keys]})
    s_in = [_nanmean(v) for k, v in per_image.items() if k[1] == "in"]
    s_out = [_nanmean(v) for k, v in per_image.items() if k[1] == "out"]
    return df, df_mean, _auroc(s_in, s_out)
