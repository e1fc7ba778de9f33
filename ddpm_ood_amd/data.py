"""Ingest for the reconstruction path: id files -> batches, plus the synthetic datasets.

The batch contract is the reference's (/root/reference/src/trainers/reconstruct.py:123,193):
``batch["image"]`` fp32 [B, C, *spatial] in [0, 1] and
``batch["image_meta_dict"]["filename_or_obj"][b]``.
Restates the parts of /root/reference/src/data/get_train_and_val_dataloader.py the path
relies on: one-row CSV of file paths (:10-16, the row is read as the header), ``first_n``
truncation before the rank split (:17-18), and the per-image ``val_transforms`` in the reference's
order (:67-84): load, EnsureChannelFirst + ``x[0, None]`` for grayscale data (a 4-D BraTS NIfTI is
channel-LAST on disk), CenterSpatialCrop (:61-65), area Resize (:55-59), min-max ScaleIntensity to
[0, 1] (:76), v/h flips (:77-82) -- all BEFORE batching, so id lists whose volumes differ in native
shape work; then the rank partition (:21-31 -- here a round-robin split without padding
duplicates).  File formats:
``.npy`` (the reference's computer-vision datasets), single-file NIfTI-1 ``.nii`` / ``.nii.gz`` (its
Medical-Decathlon volumes; read here with numpy as nibabel's ``get_fdata`` would: Fortran order,
``scl_slope`` / ``scl_inter`` applied, no reorientation -- what MONAI's ``LoadImage`` hands on), ``.npz``
archives and synthetic specs; of the PIL formats MONAI's LoadImage reads: PNG (8 / 16-bit grey, grey + alpha, RGB, RGBA,
non-interlaced) and binary PGM / PPM, with PILReader's axis swap -- JPEG / TIFF are not read.

Synthetic id specs (no dataset can be downloaded here):
    synthetic:<kind>[:n=N][:size=S][:channels=C][:seed=K][:mix=P][:name=X]
        kind in {blobs, noise, speckle (blobs + P % uniform noise), blobs3d, noise3d}
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


def synthetic_images(kind: str, n: int, channels: int = 1, size: int = 32, seed: int = 0, mix: int = 10) -> torch.Tensor:
    """kind "speckle": blobs with ``mix`` percent of uniform noise blended in -- a near-distribution OOD set whose
    AUROC against plain blobs is neither 0.5 nor saturated (used to make the AUROC checks sensitive)."""
    gen = torch.Generator().manual_seed(seed)
    if kind == "blobs3d":
        return scale_intensity(_blobs3d(n, channels, size, gen))
    if kind == "noise3d":
        return scale_intensity(torch.rand(n, channels, size, size, size, generator=gen))
    if kind == "blobs":
        x = _blobs(n, channels, size, gen)
    elif kind == "speckle":
        x = scale_intensity(_blobs(n, channels, size, gen))
        x = (1 - mix / 100.0) * x + (mix / 100.0) * torch.rand(n, channels, size, size, generator=gen)
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
    kw = {"n": 64, "size": 32, "channels": 1, "seed": 0, "mix": 10}
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


def channel_first(a: torch.Tensor, path: str, is_grayscale: bool, spatial_dimension: int) -> torch.Tensor:
    """``EnsureChannelFirstd`` + ``x[0, None]`` of the reference's grayscale pipeline
    (/root/reference/src/data/get_train_and_val_dataloader.py:69-73), per image.

    MONAI 1.2's readers decide where the channel is: a NIfTI keeps at most three spatial axes, any further
    axis is the channel (BraTS: X x Y x Z x 4 -> channel LAST on disk); a ``.npy`` array carries no channel
    information, so it is taken as channel-less.  Colour data (``is_grayscale=0``) gets no EnsureChannelFirst
    in the reference: the file has to be channel-first already.  Anything else is ambiguous and raises."""
    nifti = path.endswith((".nii", ".nii.gz"))
    pil = path.lower().endswith(PIL_SUFFIXES)
    if is_grayscale:
        if nifti and a.ndim == 4 and spatial_dimension == 3:
            a = a.movedim(-1, 0)
        elif pil and a.ndim == 3 and spatial_dimension == 2:
            a = a.movedim(-1, 0)  # colour image file: PILReader marks the last axis as the channel, x[0, None] keeps R
        elif a.ndim == spatial_dimension:
            a = a[None]
        else:
            raise ValueError(f"{path}: array of shape {tuple(a.shape)} is ambiguous for --is_grayscale=1 "
                             f"--spatial_dimension={spatial_dimension}: expected {spatial_dimension} spatial axes"
                             + (" (or a 4-D NIfTI whose last axis is the modality)" if spatial_dimension == 3 else ""))
        return a[0, None, ...]
    if a.ndim != spatial_dimension + 1:
        raise ValueError(f"{path}: array of shape {tuple(a.shape)} is not channel-first [C, *spatial{spatial_dimension}] "
                         f"(--is_grayscale=0 applies no EnsureChannelFirst, as in the reference)")
    return a


def center_crop(a: torch.Tensor, image_roi) -> torch.Tensor:
    """monai ``CenterSpatialCrop`` on [C, *spatial]: roi entries <= 0 keep the axis, larger-than-image entries are
    clipped, start = size // 2 - roi // 2 (NOT (size - roi) // 2: they differ for even size / odd roi)."""
    sp = a.shape[1:]
    if len(image_roi) != len(sp):
        raise ValueError(f"--image_roi {tuple(image_roi)} does not match the {len(sp)} spatial axes")
    sl = [slice(None)]
    for r, s in zip(image_roi, sp):
        r = min(int(r), s) if int(r) > 0 else s
        start = max(s // 2 - r // 2, 0)
        sl.append(slice(start, start + r))
    return a[tuple(sl)]


def transform_image(a: torch.Tensor, image_roi=None, image_size=None, add_vflip=False, add_hflip=False) -> torch.Tensor:
    """The reference's per-image ``val_transforms`` after the channel handling, in its order
    (get_train_and_val_dataloader.py:74-82): centre crop, area resize, min-max scale to [0, 1], flips of the
    first / second spatial axis.  a: fp32 [C, *spatial]."""
    if image_roi:
        a = center_crop(a, image_roi)
    if image_size:
        a = F.interpolate(a[None], size=(int(image_size),) * (a.ndim - 1), mode="area")[0]
    a = scale_intensity(a[None])[0]
    if add_vflip:
        a = torch.flip(a, dims=(1,))
    if add_hflip:
        a = torch.flip(a, dims=(2,))
    return a.contiguous()


def _png_unfilter(raw: bytes, height: int, stride: int, bpp: int) -> np.ndarray:
    """PNG scanline filters 0..4 (None, Sub, Up, Average, Paeth; PNG specification section 9) -> [height, stride] uint8.
    None / Up are whole-row numpy operations, Sub is a per-channel running sum (mod 256); Average and Paeth depend on the
    reconstructed left neighbour and run per byte -- over Python ints of a bytearray (about ten times faster than indexing
    numpy scalars: 0.03 s instead of 0.3-1 s for a 256 x 256 RGB image)."""
    out = np.zeros((height, stride), dtype=np.uint8)
    prev = bytes(stride)
    pos = 0
    for y in range(height):
        ft = raw[pos]
        line = raw[pos + 1:pos + 1 + stride]
        pos += stride + 1
        if ft == 0:
            cur = bytes(line)
        elif ft == 2:
            cur = ((np.frombuffer(line, dtype=np.uint8).astype(np.uint16) + np.frombuffer(prev, dtype=np.uint8)) & 255) \
                .astype(np.uint8).tobytes()
        elif ft == 1:
            px = np.frombuffer(line, dtype=np.uint8).astype(np.uint32)
            n = stride // bpp
            head = np.cumsum(px[:n * bpp].reshape(n, bpp), axis=0, dtype=np.uint32) & 255
            cur = head.astype(np.uint8).tobytes()
            if n * bpp != stride:  # (never for the colour types read_png accepts: stride = width x bpp)
                raise ValueError("PNG: row length is not a multiple of the pixel size")
        elif ft in (3, 4):
            c = bytearray(line)
            if ft == 3:
                for x in range(stride):
                    left = c[x - bpp] if x >= bpp else 0
                    c[x] = (c[x] + ((left + prev[x]) >> 1)) & 255
            else:
                for x in range(stride):
                    left = c[x - bpp] if x >= bpp else 0
                    up = prev[x]
                    ul = prev[x - bpp] if x >= bpp else 0
                    pa, pb, pc = abs(up - ul), abs(left - ul), abs(left + up - 2 * ul)
                    c[x] = (c[x] + (left if pa <= pb and pa <= pc else up if pb <= pc else ul)) & 255
            cur = bytes(c)
        else:
            raise ValueError(f"PNG: unknown filter type {ft}")
        out[y] = np.frombuffer(cur, dtype=np.uint8)
        prev = cur
    return out


def read_png(path: str) -> np.ndarray:
    """Minimal PNG reader (zlib + the five scanline filters): 8 / 16-bit greyscale, greyscale + alpha, RGB, RGBA,
    non-interlaced -- what the reference's ``LoadImaged`` gets from PIL for such files.  -> [H, W] or [H, W, C] float32."""
    import struct
    import zlib

    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if hdr is None:
        raise ValueError(f"{path}: PNG without IHDR")
    w, h, depth, ctype, _, _, interlace = hdr
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if channels is None or depth not in (8, 16) or interlace:
        raise NotImplementedError(f"{path}: PNG colour type {ctype} / bit depth {depth} / interlace {interlace} is not read "
                                  "(greyscale, greyscale + alpha, RGB, RGBA at 8 or 16 bits, non-interlaced are)")
    if depth == 16 and channels >= 3:
        # PIL (what MONAI's LoadImage sees, reference src/data/get_train_and_val_dataloader.py:60-76) converts 16-bit RGB(A)
        # files to 8 bits per channel on open; returning the 16-bit values here would silently differ from the reference
        raise NotImplementedError(f"{path}: 16-bit RGB / RGBA PNG -- PIL truncates these to 8 bits per channel; convert the file "
                                  "to 8-bit (or to .npy) so that both pipelines see the same numbers")
    bpp = channels * depth // 8
    rows = _png_unfilter(zlib.decompress(b"".join(idat)), h, w * bpp, bpp)
    a = rows.reshape(h, w, channels, depth // 8)
    a = a[..., 0].astype(np.float32) if depth == 8 else (a[..., 0].astype(np.float32) * 256 + a[..., 1])
    return a[..., 0] if channels == 1 else a


def read_pnm(path: str) -> np.ndarray:
    """Binary PGM (P5) / PPM (P6) -> [H, W] or [H, W, 3] float32."""
    data = open(path, "rb").read()
    toks, pos = [], 0
    while len(toks) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        toks.append(data[pos:end])
        pos = end
    magic, w, h, maxv = toks[0], int(toks[1]), int(toks[2]), int(toks[3])
    if magic not in (b"P5", b"P6"):
        raise NotImplementedError(f"{path}: only binary PGM (P5) / PPM (P6) are read")
    c = 1 if magic == b"P5" else 3
    dt = np.dtype(">u2") if maxv > 255 else np.uint8
    a = np.frombuffer(data, dtype=dt, count=w * h * c, offset=pos + 1).reshape(h, w, c).astype(np.float32)
    return a[..., 0] if c == 1 else a


PIL_SUFFIXES = (".png", ".pgm", ".ppm")


def read_image(path: str) -> torch.Tensor:
    if path.endswith((".nii", ".nii.gz")):
        return torch.from_numpy(read_nifti(path))
    if path.endswith(".npy"):
        return torch.from_numpy(np.load(path).astype(np.float32))
    if path.lower().endswith(PIL_SUFFIXES):
        # MONAI's PILReader (reverse_indexing=True, its default) swaps the first two axes of what PIL returns, so that the
        # array is [W, H(, C)] like the other readers' -- SURVEY Appendix A, recalled; the channel stays LAST
        a = read_png(path) if path.lower().endswith(".png") else read_pnm(path)
        return torch.from_numpy(np.ascontiguousarray(np.swapaxes(a, 0, 1)))
    raise NotImplementedError(f"{path}: .npy, NIfTI-1 (.nii / .nii.gz), .png and binary .pgm / .ppm files are ingested; "
                              "other PIL formats (JPEG, TIFF, ...) are not")


def load_ids(ids: str, is_grayscale: bool = False, first_n=None, spatial_dimension: int = 2, **transform):
    """-> (images: list of transformed fp32 [C, *spatial] tensors, names list[str]).

    Every image goes through the whole per-image pipeline (channel handling, crop, resize, scale, flips) BEFORE
    anything is batched, like the reference's CacheDataset of transformed items: volumes of different native
    shapes (the Decathlon out-sets) are fine as long as crop / resize bring them to one shape, and only the
    transformed size is kept in memory."""
    ids = str(ids)
    if ids.startswith("synthetic:"):
        kind, kw = _parse_spec(ids)
        n = kw["n"] if not first_n else min(kw["n"], int(first_n))
        x = synthetic_images(kind, kw["n"], kw["channels"], kw["size"], kw["seed"], kw["mix"])[:n]
        return [transform_image(a, **transform) for a in x], [f"{kind}_{kw['seed']}_{i:06d}.npy" for i in range(n)]
    p = Path(ids)
    if not p.exists():
        raise FileNotFoundError(f"Cannot find id file {p}")
    if p.suffix == ".npz":
        z = np.load(p, allow_pickle=False)
        x = torch.from_numpy(np.asarray(z["images"], dtype=np.float32))
        names = [str(s) for s in z["names"]] if "names" in z else [f"{p.stem}_{i:06d}.npy" for i in range(len(x))]
        if first_n:
            x, names = x[: int(first_n)], names[: int(first_n)]
        return [transform_image(a, **transform) for a in x], names
    # reference format: a CSV whose single (header) row lists the image files
    with open(p, "r") as f:
        row = [s.strip() for s in f.readline().strip().split(",") if s.strip()]
    if first_n:
        row = row[: int(first_n)]
    imgs = [transform_image(channel_first(read_image(path), path, is_grayscale, spatial_dimension), **transform)
            for path in row]
    return imgs, row


class ListLoader:
    """Minimal stand-in for monai ThreadDataLoader over a cached, transformed dataset."""

    def __init__(self, images, names: List[str], batch_size: int, drop_last: bool = False,
                 indices: List[int] = None, all_names: List[str] = None):
        # images: one [N, C, *spatial] tensor, or a list of [C, *spatial] tensors when the transformed shapes
        # differ (then a batch is stacked on demand and must be homogeneous, like torch's default collate)
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
            image = self.images[s:e]
            if isinstance(image, list):
                if len({tuple(a.shape) for a in image}) != 1:
                    raise RuntimeError(f"cannot batch images of different shapes {[tuple(a.shape) for a in image]}: "
                                       "pass --image_roi / --image_size, or --batch_size=1")
                image = torch.stack(image)
            yield {"image": image, "image_meta_dict": {"filename_or_obj": self.names[s:e]},
                   "index": self.indices[s:e]}


def partition(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin shard of the (first_n-truncated) list; no padding."""
    return list(range(rank, n_items, world))


def get_data_loader(ids: str, batch_size: int, first_n=None, is_grayscale: bool = False, image_size=None,
                    add_vflip: bool = False, add_hflip: bool = False, drop_last: bool = False,
                    spatial_dimension: int = 2, image_roi=None, rank: int = 0, world: int = 1) -> ListLoader:
    imgs, names = load_ids(ids, is_grayscale=is_grayscale, first_n=first_n, spatial_dimension=spatial_dimension,
                           image_roi=image_roi, image_size=image_size, add_vflip=add_vflip, add_hflip=add_hflip)
    for a, n in zip(imgs, names):
        if a.ndim != 1 + spatial_dimension:
            raise ValueError(f"--spatial_dimension={spatial_dimension} but {n} is {tuple(a.shape)}")
    print(f"Found {len(names)} subjects.")
    idx = partition(len(names), rank, world)
    mine = [imgs[i] for i in idx]
    if len({tuple(a.shape) for a in mine}) <= 1:
        c = 1 if is_grayscale else 3
        mine = torch.stack(mine) if mine else torch.zeros((0, c) + (0,) * spatial_dimension)
    return ListLoader(mine, [names[i] for i in idx], batch_size, drop_last, indices=idx, all_names=names)
