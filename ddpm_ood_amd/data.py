"""Ingest for the reconstruction path: id files -> batches, plus the synthetic datasets.

The batch contract is the reference's (/root/reference/src/trainers/reconstruct.py:123,193):
``batch["image"]`` fp32 [B, C, *spatial] in [0, 1] and
``batch["image_meta_dict"]["filename_or_obj"][b]``.
Restates the parts of /root/reference/src/data/get_train_and_val_dataloader.py the path
relies on: one-row CSV of file paths (:10-16, the row is read as the header), ``first_n``
truncation before the rank split (:17-18), per-image min-max ScaleIntensity to [0, 1] (:76),
optional area resize (:55-59), v/h flip variants (:77-82), rank partition (:21-31 -- here a
round-robin split without padding duplicates).  File formats:
``.npy`` (the reference's computer-vision datasets), single-file NIfTI-1 ``.nii`` / ``.nii.gz`` (its
Medical-Decathlon volumes; read here with numpy as nibabel's ``get_fdata`` would: Fortran order,
``scl_slope`` / ``scl_inter`` applied, no reorientation -- what MONAI's ``LoadImage`` hands on), ``.npz``
archives and synthetic specs.  PIL formats are not read.

Synthetic id specs (no dataset can be downloaded here):
    synthetic:<kind>[:n=N][:size=S][:channels=C][:seed=K]     kind in {blobs, noise, blobs3d, noise3d}
3-D volumes ([N, C, D, H, W]; ``.npy`` files of shape (D, H, W) or (C, D, H, W)) are handled with the
same transforms (crop, area resize, min-max scale, flips of the first / second spatial axis).
"""

from __future__ import annotations

from pathlib import Path
from typing import Iterator, List

import numpy as np
import torch
import torch.nn.functional as F


def _blobs(n: int, channels: int, size: int, gen: torch.Generator) -> torch.Tensor:
    """In-distribution images: a sum of 3-5 random Gaussian blobs, min-max scaled."""
    ys, xs = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32),
                            indexing="ij")
    out = torch.zeros(n, channels, size, size)
    for i in range(n):
        k = int(torch.randint(3, 6, (1,), generator=gen))
        for c in range(channels):
            img = torch.zeros(size, size)
            for _ in range(k):
                cy, cx = (torch.rand(2, generator=gen) * size).tolist()
                sig = float(torch.rand(1, generator=gen)) * size / 6 + size / 16
                amp = float(torch.rand(1, generator=gen)) * 0.8 + 0.2
                img += amp * torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (2 * sig * sig))
            out[i, c] = img
    return out


def _blobs3d(n: int, channels: int, size: int, gen: torch.Generator) -> torch.Tensor:
    g = torch.arange(size, dtype=torch.float32)
    zs, ys, xs = torch.meshgrid(g, g, g, indexing="ij")
    out = torch.zeros(n, channels, size, size, size)
    for i in range(n):
        k = int(torch.randint(3, 6, (1,), generator=gen))
        for c in range(channels):
            for _ in range(k):
                cz, cy, cx = (torch.rand(3, generator=gen) * size).tolist()
                sig = float(torch.rand(1, generator=gen)) * size / 6 + size / 16
                amp = float(torch.rand(1, generator=gen)) * 0.8 + 0.2
                out[i, c] += amp * torch.exp(-((zs - cz) ** 2 + (ys - cy) ** 2 + (xs - cx) ** 2) / (2 * sig * sig))
    return out


def synthetic_images(kind: str, n: int, channels: int = 1, size: int = 32, seed: int = 0) -> torch.Tensor:
    gen = torch.Generator().manual_seed(seed)
    if kind == "blobs3d":
        return scale_intensity(_blobs3d(n, channels, size, gen))
    if kind == "noise3d":
        return scale_intensity(torch.rand(n, channels, size, size, size, generator=gen))
    if kind == "blobs":
        x = _blobs(n, channels, size, gen)
    elif kind == "noise":
        x = torch.rand(n, channels, size, size, generator=gen)
    else:
        raise ValueError(f"unknown synthetic kind {kind}")
    return scale_intensity(x)


def scale_intensity(x: torch.Tensor) -> torch.Tensor:
    """monai ScaleIntensity(minv=0, maxv=1) per image (constant images map to 0)."""
    flat = x.reshape(x.shape[0], -1)
    mn = flat.min(dim=1).values.reshape(-1, *([1] * (x.ndim - 1)))
    mx = flat.max(dim=1).values.reshape(-1, *([1] * (x.ndim - 1)))
    rng = mx - mn
    return torch.where(rng > 0, (x - mn) / torch.where(rng > 0, rng, torch.ones_like(rng)), torch.zeros_like(x))


def _parse_spec(spec: str):
    parts = spec.split(":")
    kind = parts[1]
    kw = {"n": 64, "size": 32, "channels": 1, "seed": 0}
    for p in parts[2:]:
        k, v = p.split("=")
        if k != "name":  # name= only labels the results file (trainer.dataset_stem)
            kw[k] = int(v)
    return kind, kw


_NIFTI_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}


def read_nifti(path) -> np.ndarray:
    """Single-file NIfTI-1 (.nii / .nii.gz) -> float32 array of shape dim[1..ndim] (x fastest on disk)."""
    import gzip

    path = str(path)
    with (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")) as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError(f"{path}: too short for a NIfTI-1 header")
    for order in ("<", ">"):
        if int(np.frombuffer(raw, order + "i4", 1, 0)[0]) == 348:
            break
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr != 348)")
    if raw[344:347] not in (b"n+1",):
        raise ValueError(f"{path}: only single-file NIfTI-1 ('n+1') is read, magic = {raw[344:348]!r}")
    dim = np.frombuffer(raw, order + "i2", 8, 40)
    ndim = int(dim[0])
    if not 1 <= ndim <= 7:
        raise ValueError(f"{path}: bad dim[0] = {ndim}")
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    while len(shape) > 1 and shape[-1] == 1:  # trailing singleton axes (nibabel keeps them; LoadImage squeezes)
        shape = shape[:-1]
    code = int(np.frombuffer(raw, order + "i2", 1, 70)[0])
    if code not in _NIFTI_DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {code}")
    vox_offset = int(np.frombuffer(raw, order + "f4", 1, 108)[0])
    slope, inter = (float(v) for v in np.frombuffer(raw, order + "f4", 2, 112))
    n = int(np.prod(shape))
    data = np.frombuffer(raw, order + _NIFTI_DTYPES[code], n, max(vox_offset, 352)).reshape(shape, order="F")
    data = data.astype(np.float32)
    if slope != 0.0 and np.isfinite(slope) and np.isfinite(inter) and (slope != 1.0 or inter != 0.0):
        data = data * np.float32(slope) + np.float32(inter)
    return np.ascontiguousarray(data)


def load_ids(ids: str, is_grayscale: bool = False, first_n=None, spatial_dimension: int = 2):
    """-> (images fp32 [N, C, *spatial] unscaled, names list[str])."""
    ids = str(ids)
    if ids.startswith("synthetic:"):
        kind, kw = _parse_spec(ids)
        n = kw["n"] if not first_n else min(kw["n"], int(first_n))
        x = synthetic_images(kind, kw["n"], kw["channels"], kw["size"], kw["seed"])[:n]
        return x, [f"{kind}_{kw['seed']}_{i:06d}.npy" for i in range(n)]
    p = Path(ids)
    if not p.exists():
        raise FileNotFoundError(f"Cannot find id file {p}")
    if p.suffix == ".npz":
        z = np.load(p, allow_pickle=False)
        x = torch.from_numpy(np.asarray(z["images"], dtype=np.float32))
        names = [str(s) for s in z["names"]] if "names" in z else [f"{p.stem}_{i:06d}.npy" for i in range(len(x))]
        if first_n:
            x, names = x[: int(first_n)], names[: int(first_n)]
        return x, names
    # reference format: a CSV whose single (header) row lists the image files
    with open(p, "r") as f:
        row = [s.strip() for s in f.readline().strip().split(",") if s.strip()]
    if first_n:
        row = row[: int(first_n)]
    imgs = []
    for path in row:
        if path.endswith((".nii", ".nii.gz")):
            a = torch.from_numpy(read_nifti(path))
        elif path.endswith(".npy"):
            a = torch.from_numpy(np.load(path).astype(np.float32))
        else:
            raise NotImplementedError(f"{path}: .npy and NIfTI-1 (.nii / .nii.gz) files are ingested; PIL formats are not")
        if a.ndim == spatial_dimension:
            a = a[None]
        if is_grayscale:
            a = a[0, None, ...]
        imgs.append(a)
    return torch.stack(imgs), row


class ListLoader:
    """Minimal stand-in for monai ThreadDataLoader over a cached, transformed dataset."""

    def __init__(self, images: torch.Tensor, names: List[str], batch_size: int, drop_last: bool = False,
                 indices: List[int] = None, all_names: List[str] = None):
        self.images, self.names, self.batch_size, self.drop_last = images, names, batch_size, drop_last
        self.indices = list(range(len(names))) if indices is None else list(indices)  # global image ids
        self.all_names = list(names) if all_names is None else list(all_names)

    def __len__(self):
        n = len(self.names)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[dict]:
        n = len(self.names)
        for s in range(0, n, self.batch_size):
            e = min(n, s + self.batch_size)
            if self.drop_last and e - s < self.batch_size:
                return
            yield {"image": self.images[s:e], "image_meta_dict": {"filename_or_obj": self.names[s:e]},
                   "index": self.indices[s:e]}


def partition(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin shard of the (first_n-truncated) list; no padding."""
    return list(range(rank, n_items, world))


def get_data_loader(ids: str, batch_size: int, first_n=None, is_grayscale: bool = False, image_size=None,
                    add_vflip: bool = False, add_hflip: bool = False, drop_last: bool = False,
                    spatial_dimension: int = 2, image_roi=None, rank: int = 0, world: int = 1) -> ListLoader:
    x, names = load_ids(ids, is_grayscale=is_grayscale, first_n=first_n, spatial_dimension=spatial_dimension)
    if x.ndim != 2 + spatial_dimension:
        raise ValueError(f"--spatial_dimension={spatial_dimension} but the images are {tuple(x.shape[1:])}")
    print(f"Found {len(names)} subjects.")
    if image_roi:
        roi = [min(r, s) if r > 0 else s for r, s in zip(image_roi, x.shape[2:])]
        sl = tuple(slice((s - r) // 2, (s - r) // 2 + r) for r, s in zip(roi, x.shape[2:]))
        x = x[(slice(None), slice(None)) + sl]
    if image_size:
        x = F.interpolate(x, size=(int(image_size),) * spatial_dimension, mode="area")
    x = scale_intensity(x)
    if add_vflip:
        x = torch.flip(x, dims=(2,))
    if add_hflip:
        x = torch.flip(x, dims=(3,))
    idx = partition(len(names), rank, world)
    return ListLoader(x[idx].contiguous(), [names[i] for i in idx], batch_size, drop_last, indices=idx,
                      all_names=names)
