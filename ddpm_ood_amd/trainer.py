"""``Reconstruct``: the multi-t reconstruction trainer with the reference's surface.

Mirrors /root/reference/src/trainers/base.py:19-164 (setup half: device / process group,
stage-1 model, ``small`` / ``big`` UNet constructor arguments, schedule parameters, checkpoint
load with the reference's exceptions) and /root/reference/src/trainers/reconstruct.py:29-330
(``get_scores`` hot loops, CSV files, ``_vflip`` / ``_hflip`` out-set variants).

What is MI355X-native here (and differs in mechanism, not in results, from the reference):
  * the UNet forward is one native call (ddpm_unet_forward), the PLMS update / add_noise /
    clamp+MSE are single fused HIP kernels; there is no autocast (fp32 throughout, Q5);
  * timesteps live on the device (one cached int64 tensor per step value, Q18) and scores
    come back with ONE device->host copy per batch instead of 2*B ``.item()`` syncs per t;
  * noise is an explicit, host-generated, per-image seeded input (``--seed`` finally has an
    effect, Q2), so results do not depend on batch composition or rank count;
  * images are sharded round-robin over ranks and scores return through a single RCCL
    all_gather of a dense [ceil(n / world), n_t * 2 + 1] fp32 tensor per rank (ids ride in column 0; the
    shard capacity is static, so no size exchange precedes it) instead of all_gather_object of pickled
    dict rows; only rank 0 writes the CSV (Q6).
"""

from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

import pandas as pd
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib, ops
from .data import get_data_loader
from .perceptual import PerceptualLoss
from .scheduler import DDPMScheduler, PNDMScheduler
from .unet import DiffusionModelUNet
from .vqvae import VQVAE, PassthroughVQVAE

MODEL_CONFIGS = {  # /root/reference/src/trainers/base.py:65-86
    "small": dict(num_channels=(128, 256, 256), attention_levels=(False, False, True), num_res_blocks=1,
                  num_head_channels=256),
    "big": dict(num_channels=(256, 512, 768), attention_levels=(True, True, True), num_res_blocks=2,
                num_head_channels=256),
}


def image_noise(seed: int, index: int, t_start: int, shape) -> torch.Tensor:
    """The noise input of one (image, t_start) reconstruction: host fp32 N(0, 1), a pure
    function of (seed, global image index, t_start)."""
    g = torch.Generator().manual_seed((int(seed) * 1_000_003 + int(index)) * 1_009 + int(t_start))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)


def batch_noise(seed: int, indices, t_start: int, shape) -> torch.Tensor:
    return torch.stack([image_noise(seed, i, t_start, shape[1:]) for i in indices])


def snr_shift_tables(scheduler, snr_shift: float) -> None:
    """/root/reference/src/trainers/reconstruct.py:106-117 (same code at base.py:104-116)."""
    snr = scheduler.alphas_cumprod / (1 - scheduler.alphas_cumprod)
    target_snr = snr * snr_shift
    new_alphas_cumprod = 1 / (torch.pow(target_snr, -1) + 1)
    new_alphas = torch.zeros_like(new_alphas_cumprod)
    new_alphas[0] = new_alphas_cumprod[0]
    for i in range(1, len(new_alphas)):
        new_alphas[i] = new_alphas_cumprod[i] / new_alphas_cumprod[i - 1]
    scheduler.betas = 1 - new_alphas
    scheduler.alphas = new_alphas
    scheduler.alphas_cumprod = new_alphas_cumprod


def gather_scores(ids: torch.Tensor, scores: torch.Tensor, n_max: int = None):
    """ONE all_gather of dense per-image scores (RCCL over xGMI; gloo in the CPU tests).

    ids: int32 [n_local] global image indices; scores: fp32 [n_local, n_t, 2]; n_max: the static shard capacity
    ceil(n_images / world) -- every rank can compute it from the id list, so no size exchange is needed.
    Returns (ids, scores, counts) concatenated rank-major on every rank (the reference's all_gather_object
    semantics, reconstruct.py:238-242, in 2 MB instead of 30 MB of pickles).  Short shards are padded with
    id = -1; the padding is dropped after the gather and ``counts`` (rows per rank) is read off the ids."""
    if not dist.is_initialized():
        return ids, scores, [int(ids.shape[0])]
    world = dist.get_world_size()  # a 1-rank group still goes through the collective (same code path as N ranks)
    if n_max is None:
        n_max = int(ids.shape[0]) if world == 1 else None
    if n_max is None or ids.shape[0] > n_max:
        raise ValueError(f"gather_scores: shard of {ids.shape[0]} rows does not fit the static capacity {n_max}")
    n_t = scores.shape[1]
    # ids travel inside the same fp32 payload (exact for ids < 2^24): one collective, not two
    if ids.numel() and int(ids.max()) >= 1 << 24:
        raise ValueError(f"gather_scores: image id {int(ids.max())} is not exact in the fp32 payload (ids must be < 2^24)")
    payload = torch.full((n_max, n_t * 2 + 1), -1.0, dtype=torch.float32, device=scores.device)
    payload[: ids.shape[0], 0] = ids.to(torch.float32)
    payload[: ids.shape[0], 1:] = scores.reshape(ids.shape[0], n_t * 2)  # (an empty shard has 0 rows)
    dev = scores.device
    if dist.get_backend() == "gloo" and payload.is_cuda:  # test hook (2 ranks on one GPU): gloo moves host memory
        payload = payload.cpu()
    out = torch.empty((world * n_max, n_t * 2 + 1), dtype=torch.float32, device=payload.device)
    dist.all_gather_into_tensor(out, payload)
    out = out.to(dev)
    valid = out[:, 0] >= 0
    counts = valid.reshape(world, n_max).sum(dim=1).tolist()
    out = out[valid]
    return out[:, 0].to(torch.int32), out[:, 1:].reshape(-1, n_t, 2), [int(c) for c in counts]


def sub_batch(batch: dict, keep):
    """The loader batch restricted to the items `keep` (the numeric guard re-runs only the images that overflowed)."""
    img = batch["image"]
    sel = torch.as_tensor(keep, dtype=torch.long)
    image = img[sel.to(img.device)] if torch.is_tensor(img) else torch.stack([img[int(i)] for i in keep])
    names = batch["image_meta_dict"]["filename_or_obj"]
    return {"image": image, "index": [batch["index"][i] for i in keep],
            "image_meta_dict": {"filename_or_obj": [names[i] for i in keep]}}


def rows_from_scores(ids, scores, counts, t_values, name_of, batch_size: int, dataset_name: str):
    """Dense gathered scores -> the reference's row dicts, in the reference's order: rank-major
    (all_gather_object, reconstruct.py:238-242), and inside a rank per batch, per t_start, per image
    (reconstruct.py:128,192-204)."""
    ids = [int(i) for i in ids]
    results = []
    start = 0
    for n_rank in counts:
        for s in range(start, start + n_rank, batch_size):
            e = min(start + n_rank, s + batch_size)
            for j, t in enumerate(t_values):
                for b in range(s, e):
                    filename = name_of.get(ids[b], str(ids[b]))
                    stem = Path(filename).stem.replace(".nii", "").replace(".gz", "")
                    results.append({"filename": stem, "type": dataset_name, "t": int(t),
                                    "perceptual_difference": float(scores[b, j, 0]), "mse": float(scores[b, j, 1])})
        start += n_rank
    return results


class BaseTrainer:
    def __init__(self, args):
        # every rank reports a missing device / library BEFORE the non-zero ranks' output is silenced
        if not torch.cuda.is_available():
            raise RuntimeError("No ROCm device visible: the HIP reconstruction path has no CPU fallback "
                               "(the CPU oracle under oracle/ is test infrastructure only)")
        _lib.load()  # fail before any work if the native library is missing
        # initialise the process group if launched with torchrun (base.py:22-33)
        if "LOCAL_RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            print("Setting up DDP.")
            self.ddp = True
            local_rank = int(os.environ["LOCAL_RANK"])
            if local_rank != 0:
                sys.stdout = sys.stderr = open(os.devnull, "w")
            if not dist.is_initialized():
                # backend "nccl" IS RCCL on ROCm: one process per GPU, xGMI underneath.  DDPM_DIST_BACKEND=gloo is the
                # test hook for boxes with fewer GPUs than ranks (RCCL refuses two ranks on one device); with
                # DDPM_DIST_SHARED_DEVICE=1 every rank then computes on cuda:0 (tests/test_gpu_dist.py)
                dist.init_process_group(backend=os.environ.get("DDPM_DIST_BACKEND", "nccl"), init_method="env://")
            shared = os.environ.get("DDPM_DIST_SHARED_DEVICE", "0") == "1"
            self.device = torch.device("cuda:0" if shared else f"cuda:{local_rank}")
        else:
            self.ddp = False
            self.device = torch.device("cuda:0")
        torch.cuda.set_device(self.device)
        self.rank = dist.get_rank() if self.ddp else 0
        self.world = dist.get_world_size() if self.ddp else 1

        print(f"Arguments: {str(args)}")
        for k, v in vars(args).items():
            print(f"  {k}: {v}")

        self._setup_stage1(args)
        self._setup_unet(args)
        self._setup_schedule(args)
        self._setup_geometry(args)
        self._restore_checkpoint(args)
        # no optimizer, no GradScaler, no DDP wrap: inference only, every rank reads the same
        # checkpoint file instead of the reference's DDP parameter broadcast (Q15)


    # ---- the pieces of the reference's BaseTrainer.__init__ (base.py:44-158) the reconstruction path needs, one loader each.
    # Attribute names, printed messages and exception types are the reference's (row a2 of SURVEY 8: a caller that catches
    # FileNotFoundError / ValueError or reads trainer.vqvae_config keeps working); the structure is not.
    @staticmethod
    def _require(path: Path, what: str) -> Path:
        if not path.exists():
            raise FileNotFoundError(f"Cannot find {what} {path}")
        return path

    def _setup_stage1(self, args):
        """Stage-1 model (base.py:44-64): a VQ-VAE checkpoint + its vqvae_config.json next to it, or the identity."""
        self.ddpm_channels = 1 if args.is_grayscale else 3
        if not args.vqvae_checkpoint:
            self.vqvae_model = PassthroughVQVAE()
            return
        ckpt = self._require(Path(args.vqvae_checkpoint), "VQ-VAE checkpoint")
        cfg = self._require(ckpt.parent / "vqvae_config.json", "VQ-VAE config")
        self.vqvae_config = json.loads(cfg.read_text())
        self.vqvae_model = VQVAE(**self.vqvae_config)
        state = torch.load(ckpt, map_location="cpu", weights_only=False)["model_state_dict"]
        self.vqvae_model.load_state_dict(state)
        self.vqvae_model.to(self.device).eval()
        print("Loaded vqvae model with config:")
        for k, v in self.vqvae_config.items():
            print(f"  {k}: {v}")
        self.ddpm_channels = self.vqvae_config["embedding_dim"]

    def _setup_unet(self, args):
        """The denoiser (base.py:65-86): `small` / `big` channel tables, one channel count on both sides."""
        if args.model_type not in MODEL_CONFIGS:
            raise ValueError(f"Do not recognise model type {args.model_type}")
        self.model = DiffusionModelUNet(spatial_dims=args.spatial_dimension, in_channels=self.ddpm_channels,
                                        out_channels=self.ddpm_channels, with_conditioning=False,
                                        use_proj_attn=bool(getattr(args, "use_proj_attn", 0)),
                                        **MODEL_CONFIGS[args.model_type]).to(self.device)
        print(f"{sum(p.numel() for p in self.model.parameters()):,} model parameters")

    def _setup_schedule(self, args):
        """Noise schedule attributes + the training scheduler (base.py:88-117); --snr_shift rewrites its tables."""
        for name in ("prediction_type", "beta_schedule", "beta_start", "beta_end", "b_scale", "snr_shift"):
            setattr(self, name, getattr(args, name))
        self.scheduler = DDPMScheduler(num_train_timesteps=1000, prediction_type=self.prediction_type,
                                       schedule=self.beta_schedule, beta_start=self.beta_start, beta_end=self.beta_end)
        if self.snr_shift != 1:
            print("Changing scheduler parameters to shift SNR")
            snr_shift_tables(self.scheduler, self.snr_shift)
        self.simplex_noise = bool(args.simplex_noise)
        if self.simplex_noise:
            raise NotImplementedError("--simplex_noise is off the path (default 0, in no BASELINE config)")

    def _setup_geometry(self, args):
        """Spatial bookkeeping (base.py:118-131): dimension, resize target, latent padding and its inverse."""
        self.spatial_dimension = args.spatial_dimension
        self.image_size = int(args.image_size) if args.image_size else args.image_size
        self.do_latent_pad = bool(args.latent_pad)
        if self.do_latent_pad:
            self.latent_pad = args.latent_pad
            self.inverse_latent_pad = [-x for x in self.latent_pad]

    def _restore_checkpoint(self, args):
        """DDPM checkpoint (base.py:133-158; map_location added so a CPU-saved file loads, Q4): counters default to a fresh run."""
        self.run_dir = Path(args.output_dir) / args.model_name
        stem = f"checkpoint_{int(args.ddpm_checkpoint_epoch)}" if args.ddpm_checkpoint_epoch else "checkpoint"
        checkpoint_path = self.run_dir / f"{stem}.pth"
        self.start_epoch, self.best_loss, self.global_step, self.optimizer_state = 0, 1000, 0, None
        self.found_checkpoint = checkpoint_path.exists()
        if not self.found_checkpoint:
            return
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(checkpoint["model_state_dict"])
        self.start_epoch = checkpoint["epoch"] + 1
        self.global_step = checkpoint["global_step"]
        self.best_loss = checkpoint["best_loss"]
        self.optimizer_state = checkpoint.get("optimizer_state_dict")  # used by DDPMTrainer only
        print(f"Resuming training using checkpoint {checkpoint_path} at epoch {self.start_epoch}")


class Reconstruct(BaseTrainer):
    def __init__(self, args):
        super().__init__(args)
        if not self.found_checkpoint:
            raise FileNotFoundError("Failed to find a saved model checkpoint.")
        self.out_dir = self.run_dir / "ood"
        if self.rank == 0:
            self.out_dir.mkdir(exist_ok=True)
        self.seed = int(args.seed)
        self.num_inference_steps = int(getattr(args, "honour_num_inference_steps", 0)
                                       and args.num_inference_steps) or 100  # Q1: 100 is hard-coded
        self.reset_scheduler_per_t = bool(getattr(args, "reset_scheduler_per_t", 0))
        self.timestep_list = getattr(args, "timestep_list", "monai")
        # test hook (not a CLI flag): only the t-starts <= max_t_start of the reference's list are run -- a PREFIX of the
        # chained list, so the trajectories that do run are exactly the reference's (tests price the CPU oracle per forward)
        self.max_t_start = getattr(args, "max_t_start", None)
        # test hook: keep only these members of the chained t-start list (in the list's order): long chains at a few start points
        self.t_start_subset = getattr(args, "t_start_subset", None)
        self.lpips_weights = getattr(args, "lpips_weights", None)
        self._loader_args = dict(batch_size=args.batch_size, is_grayscale=bool(args.is_grayscale),
                                 image_size=self.image_size, drop_last=bool(args.drop_last),
                                 spatial_dimension=args.spatial_dimension, image_roi=args.image_roi,
                                 rank=self.rank, world=self.world)
        self.val_loader = get_data_loader(args.validation_ids,
                                          first_n=int(args.first_n_val) if args.first_n_val else args.first_n_val,
                                          **self._loader_args)
        self.in_loader = get_data_loader(args.in_ids, first_n=int(args.first_n) if args.first_n else args.first_n,
                                         **self._loader_args)
        self._ts_cache = {}
        self._pl = None
        self.last_stats = {}
        self.profile_first_steps = False  # bench.py: hipEvent-bracket the first UNet step of each t-start

    # ---- helpers -------------------------------------------------------------------------------------
    def _timesteps_tensor(self, step: int, batch: int) -> torch.Tensor:
        key = (int(step), batch)
        t = self._ts_cache.get(key)
        if t is None:
            t = torch.full((batch,), int(step), dtype=torch.int64, device=self.device)
            self._ts_cache[key] = t
        return t

    def _perceptual(self) -> PerceptualLoss:
        if self._pl is None:
            pl = PerceptualLoss(dimensions=self.spatial_dimension, include_pixel_loss=False,
                                is_fake_3d=True if self.spatial_dimension == 3 else False, lpips_normalize=True,
                                spatial=False)
            if self.lpips_weights:
                pl.perceptual_function.load_pretrained_state_dict(
                    torch.load(self.lpips_weights, map_location="cpu", weights_only=False))
            elif self.rank == 0:
                # the reference builds lpips.LPIPS(pretrained=True) (perceptual_loss.py:68-84); those weights
                # cannot be fetched here, so say loudly that this column is not LPIPS
                print("WARNING: no --lpips_weights given: LPIPS-AlexNet runs with SEEDED RANDOM weights, so the "
                      "'perceptual_difference' column (and plot_target=perceptual_difference / mse+perceptual) is "
                      "NOT an LPIPS value. Only 'mse' is comparable with the reference. Export weights on a "
                      "machine with the lpips package: torch.save(lpips.LPIPS(net='alex').state_dict(), path).",
                      file=sys.__stderr__, flush=True)
            self._pl = pl.to(self.device)
        return self._pl

    def make_scheduler(self) -> PNDMScheduler:
        s = PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, prediction_type=self.prediction_type,
                          schedule=self.beta_schedule, beta_start=self.beta_start, beta_end=self.beta_end,
                          timestep_list=self.timestep_list)
        if self.snr_shift != 1:
            snr_shift_tables(s, self.snr_shift)
        s.set_timesteps(self.num_inference_steps)
        return s

    # ---- the hot loops (reconstruct.py:72-250) ---------------------------------------------------
    @torch.no_grad()
    def get_scores(self, loader, dataset_name, inference_skip_factor):
        quiet = getattr(self, "quiet", False)
        if self.ddp and not quiet:
            sys.stdout, sys.stderr = sys.__stdout__, sys.__stderr__
            print(f"{self.rank}: {dataset_name}")
        elif not quiet:
            print(f"{dataset_name}")
        pl = self._perceptual()
        self.model.eval()
        ids_all, names_all, scores_all = [], [], []
        t_values = [int(t) for t in reversed(self.make_scheduler().timesteps)[1::inference_skip_factor]
                    if (self.max_t_start is None or int(t) <= int(self.max_t_start))
                    and (self.t_start_subset is None or int(t) in set(self.t_start_subset))]
        n_recon = n_fwd = 0
        guard = {"batches_rerun_fp32": 0, "batches_nonfinite": 0}
        _lib.status_read(clear=True)  # whatever an earlier caller left behind is not this run's
        quantises = not isinstance(self.vqvae_model, PassthroughVQVAE)
        if quantises:
            guard["vq_near_ties"] = 0  # latent positions whose two nearest codes were within 1e-5: candidates for a code flip
            _lib.vq_near_ties_read(clear=True)
        for batch in loader:
            scores, t_values, n_r, n_f, B, dt = self._score_batch(batch, pl, inference_skip_factor)
            # numeric guard (include/ddpm_ood_hip.h): a non-finite eps / reconstruction / latent set the device status word.
            # With the split-f16 kernels on, that may be an operand beyond the f16 range rather than a genuine fp32
            # overflow: run the affected IMAGES again on the fp32-MFMA kernels; what is still non-finite then is written as
            # NaN, like the reference would (reconstruct.py:188-204 has no check).
            word = _lib.status_read(clear=True)
            if word and _lib.split_f16_active():
                # only the images that came out non-finite ride again (clamp + MSE keeps a NaN: a non-finite eps, latent or
                # reconstruction ends as a non-finite score of ITS image; a per-image result does not depend on the batch it
                # rides in -- noise is a function of the image index, the PLMS history is per element)
                bad = (~torch.isfinite(scores).all(dim=2).all(dim=1)).nonzero().flatten().cpu().tolist()
                if not bad:
                    bad = list(range(B))
                print(f"WARNING: {_lib.status_text(word)} in a batch of {B} on the split-f16 kernels: running {len(bad)} image(s) "
                      f"again with fp32 MFMA products (ddpm_set_split_f16(0); permanently: DDPM_WINO44_F16X3=0 "
                      f"DDPM_CONV1X1_F16X3=0 DDPM_ATTN_F16X3=0 DDPM_DOWN_S2H=0)", file=sys.__stderr__, flush=True)
                if quantises:  # the counter must match the scores that are KEPT: a wholly discarded pass contributes nothing
                    first_pass = _lib.vq_near_ties_read(clear=True)
                    if len(bad) < B:  # (includes the re-run images' share of the first pass: an upper bound)
                        guard["vq_near_ties"] += first_pass
                prev = _lib.set_split_f16(False)
                try:
                    sub = sub_batch(batch, bad)
                    scores_b, t_values, _, n_f2, _, dt2 = self._score_batch(sub, pl, inference_skip_factor)
                    scores[torch.as_tensor(bad, device=scores.device)] = scores_b
                    n_f += n_f2
                    dt += dt2
                    word = _lib.status_read(clear=True)
                finally:
                    _lib.set_split_f16(prev)
                guard["batches_rerun_fp32"] += 1
                guard["images_rerun_fp32"] = guard.get("images_rerun_fp32", 0) + len(bad)
            if quantises:
                guard["vq_near_ties"] += _lib.vq_near_ties_read(clear=True)
            if word:
                guard["batches_nonfinite"] += 1
                print(f"WARNING: {_lib.status_text(word)} with fp32 products too: a genuine overflow of this checkpoint on "
                      f"these inputs; the affected scores are NaN / inf in the CSV, as the reference would write them",
                      file=sys.__stderr__, flush=True)
            n_recon += n_r
            n_fwd += n_f
            scores_all.append(scores)
            ids_all.append(torch.tensor(batch["index"], dtype=torch.int32))
            names_all.extend(batch["image_meta_dict"]["filename_or_obj"])
            if quiet:
                pass
            elif self.ddp:
                print(f"{self.rank}: Took {dt}s for a batch size of {B}")
            else:
                print(f"Took {dt}s for a batch size of {B}")
        self.last_stats = {"reconstructions": n_recon, "unet_forwards": n_fwd,
                           "lpips_pretrained": bool(pl.perceptual_function.pretrained), **guard}
        return self._collect(loader, dataset_name, t_values, ids_all, names_all, scores_all, quiet)

    def _score_batch(self, batch, pl, inference_skip_factor):
        """One batch through reconstruct.py:97-204: every t-start's noise, PLMS trajectory, decode, MSE and LPIPS.
        Returns (scores [B, n_t, 2] on the device, t values, reconstructions, UNet image-forwards, B, seconds)."""
        n_recon = n_fwd = 0
        sched = self.make_scheduler()  # one per batch: PLMS history leaks across t-starts (Q3)
        timesteps = sched.timesteps
        start_points = reversed(timesteps)[1::inference_skip_factor]
        if self.max_t_start is not None:
            start_points = start_points[start_points <= int(self.max_t_start)]
        if self.t_start_subset is not None:
            start_points = start_points[torch.isin(start_points, torch.as_tensor(list(self.t_start_subset), dtype=start_points.dtype))]
        t_values = [int(t) for t in start_points]

        t1 = time.time()
        images_original = batch["image"].to(self.device, non_blocking=True).float().contiguous()
        images = self.vqvae_model.encode_stage_2_inputs(images_original).float().contiguous()
        if self.do_latent_pad:
            images = F.pad(input=images, pad=self.latent_pad, mode="constant", value=0)
        B = images.shape[0]
        idx = batch["index"]
        per_t = []
        for t_start in start_points:
            if self.reset_scheduler_per_t:
                sched.set_timesteps(self.num_inference_steps)
            start_timesteps = torch.Tensor([t_start] * B).long()
            noise = batch_noise(self.seed, idx, int(t_start), images.shape).to(self.device, non_blocking=True)
            x = sched.add_noise(original_samples=images, noise=noise, timesteps=start_timesteps,
                                b_scale=self.b_scale)
            first = self.profile_first_steps
            for step in timesteps[timesteps <= t_start]:
                if first:
                    _lib.load().ddpm_prof_enable(1)
                eps = self.model(x, timesteps=self._timesteps_tensor(int(step), B))
                x, _ = sched.step(eps, step, x)
                if first:
                    _lib.load().ddpm_prof_enable(0)
                    first = False
                n_fwd += B
            if self.do_latent_pad:
                x = F.pad(input=x, pad=self.inverse_latent_pad, mode="constant", value=0).contiguous()
            prof_decode = self.profile_first_steps and not isinstance(self.vqvae_model, PassthroughVQVAE)
            if prof_decode:
                _lib.load().ddpm_prof_enable(1)
            x = self.vqvae_model.decode_stage_2_outputs(x).contiguous()
            if prof_decode:
                _lib.load().ddpm_prof_enable(0)
            mse = ops.clamp_mse_(images_original, x, self.b_scale)  # x / b_scale, clamp_(0, 1), MSE
            if self.spatial_dimension == 2:
                if images_original.shape[3] == 28:
                    pd_ = pl(F.pad(images_original, (2, 2, 2, 2)), F.pad(x, (2, 2, 2, 2)))
                else:
                    pd_ = pl(images_original, x)
                pd_ = pd_.reshape(B)
            else:
                pd_ = torch.stack([pl(images_original[b, None, ...], x[b, None, ...]).reshape(())
                                   for b in range(B)])
            per_t.append(torch.stack([pd_, mse], dim=1))
            n_recon += B
        scores = torch.stack(per_t, dim=1)  # [B, n_t, 2] on the device
        torch.cuda.current_stream().synchronize()
        t2 = time.time()
        return scores, t_values, n_recon, n_fwd, B, t2 - t1

    def _collect(self, loader, dataset_name, t_values, ids_all, names_all, scores_all, quiet):
        """reconstruct.py:238-250: everyone's rows on every rank, through ONE collective."""
        if not scores_all and not self.ddp:
            return []
        if scores_all:
            scores = torch.cat(scores_all, dim=0)
            ids = torch.cat(ids_all).to(self.device)
        else:  # a rank whose shard is empty still takes part in the collective
            scores = torch.zeros((0, len(t_values), 2), dtype=torch.float32, device=self.device)
            ids = torch.zeros((0,), dtype=torch.int32, device=self.device)
        name_of = dict(zip((int(i) for i in ids.cpu()), names_all))
        counts = [int(ids.shape[0])]
        if self.ddp:
            n_total = len(loader.all_names) if hasattr(loader, "all_names") else None
            n_max = -(-n_total // self.world) if n_total is not None else None
            if scores.is_cuda:
                torch.cuda.synchronize()  # (the batch loop already synchronised at its last status read: this prices the collective alone)
            t_g = time.perf_counter()
            ids, scores, counts = gather_scores(ids, scores, n_max)
            if scores.is_cuda:
                torch.cuda.synchronize()
            self.last_stats["gather_ms"] = round((time.perf_counter() - t_g) * 1e3, 3)  # wait for the slowest rank included
            # every rank can name every image: the id list is the same file on every rank
            name_of = {i: n for i, n in enumerate(loader.all_names)} if hasattr(loader, "all_names") else name_of
            if int(os.environ["LOCAL_RANK"]) != 0 and not quiet:
                sys.stdout = sys.stderr = open(os.devnull, "w")
        # the one device->host copy of the scores
        return rows_from_scores(ids.cpu().tolist(), scores.cpu().numpy(), counts, t_values, name_of,
                                loader.batch_size, dataset_name)

    def _write(self, results_list, name):
        if self.rank == 0:
            pd.DataFrame(results_list).to_csv(self.out_dir / f"results_{name}.csv")

    def reconstruct(self, args):
        if bool(args.run_val):
            self._write(self.get_scores(self.val_loader, "val", args.inference_skip_factor), "val")
        if bool(args.run_in):
            self._write(self.get_scores(self.in_loader, "in", args.inference_skip_factor), "in")
        if bool(args.run_out):
            for out in args.out_ids.split(","):
                print(out)
                flips = {}
                if "vflip" in out:
                    out = out.replace("_vflip", "")
                    flips, suffix = {"add_vflip": True}, "_vflip"
                elif "hflip" in out:
                    out = out.replace("_hflip", "")
                    flips, suffix = {"add_hflip": True}, "_hflip"
                else:
                    suffix = ""
                out_loader = get_data_loader(out, first_n=int(args.first_n) if args.first_n else args.first_n,
                                             **flips, **self._loader_args)
                dataset_name = dataset_stem(out) + suffix
                self._write(self.get_scores(out_loader, "out", args.inference_skip_factor), dataset_name)


def dataset_stem(ids: str) -> str:
    """Path(out).stem.split("_")[0] of the reference (reconstruct.py:287,308,327); synthetic
    specs are named by their kind."""
    if str(ids).startswith("synthetic:"):
        parts = str(ids).split(":")
        name = parts[1]
        for p in parts[2:]:
            if p.startswith("name="):
                name = p[5:]
        return name
    return Path(ids).stem.split("_")[0]
