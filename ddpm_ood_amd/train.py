"""``DDPMTrainer``: the epsilon-prediction training loop that produces the checkpoint the reconstruction path loads
(SURVEY.md 8(f) row f-3).

Mirrors /root/reference/src/trainers/ddpm_trainer.py:16-124 (epoch loop, best-loss / periodic checkpoints, the
training step: random timesteps, Gaussian noise, ``scheduler.add_noise(images * b_scale)``, MSE between the UNet
output and the noise) with Adam(lr = 2.5e-5) as at /root/reference/src/trainers/base.py:156 and the checkpoint
dict of base.py:166-187.  Differences, on purpose:
  * fp32 by default; ``--amp 1`` mirrors the reference's fp16 autocast + GradScaler
    (/root/reference/src/trainers/ddpm_trainer.py:96-109, base.py:122) over the same ATen ops;
  * since round 6 the training step of a 2-D UNet is NATIVE (``train_native.NativeUNetStep``: forward on the inference path's
    convolution kernels, backward and Adam on hand-written HIP kernels -- ddpm_conv_wgrad_f32, ddpm_gemm_f32, train_ops.hip; no
    ATen / MIOpen / rocBLAS kernel between the noisy batch and the updated parameters), and so is the 3-D latent UNet's of the LDM
    configuration (conv3d weight gradient per depth tap).  ``DDPM_TRAIN_NATIVE=0``, ``--amp 1`` and 3-D UNets without an MFMA
    tiling take the older route: PyTorch-ROCm autograd over ``unet_forward_torch``, which evaluates the SAME
    parameter holders the HIP engine reads (``DiffusionModelUNet``) with differentiable ATen ops.  Either way a checkpoint
    written here loads into the HIP inference path unchanged;
  * multi-GPU: one process per GPU; rank 0's initial parameters and buffers are broadcast once (what
    DistributedDataParallel's constructor does for the reference, base.py:160-163), gradients are averaged with ONE flat
    RCCL all_reduce per step (17.7 M parameters = 71 MB for `small`) instead of DDP's bucket hooks, and every rank
    runs the SAME number of steps per epoch on equally long shards (short shards wrap around, as
    torch.utils.data.DistributedSampler pads) -- a rank with one image less would otherwise issue one all_reduce less
    and hang the job;
  * the loss target is the noise for every --prediction_type, as in the reference (ddpm_trainer.py:99-100 regresses
    onto `noise` also under v_prediction); --augmentation / --cache_data / --num_workers are accepted and have no effect
    (the reference's two augmentation branches are identical, get_train_and_val_dataloader.py:87-91; the images are
    resident tensors here);
  * no TensorBoard / matplotlib sample grids (not installed here; off the path).
"""

from __future__ import annotations

import math
import os
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import train_ops as T
from .data import get_data_loader
from .train_native import NativeUNetStep, native_supported
from .trainer import BaseTrainer


# ---- differentiable forward over the parameter holders ------------------------------------

def _conv(holder, x, stride=1):
    c = holder.conv
    f = F.conv2d if c.weight.ndim == 4 else F.conv3d
    return f(x, c.weight, c.bias, stride=stride, padding=c.padding)


def _resnet(blk, x, emb):
    h = _conv(blk.conv1, F.silu(blk.norm1(x)))
    t = F.linear(F.silu(emb), blk.time_emb_proj.weight, blk.time_emb_proj.bias)
    h = h + t.reshape(t.shape + (1,) * (x.ndim - 2))
    h = _conv(blk.conv2, F.silu(blk.norm2(h)))
    skip = x if isinstance(blk.skip_connection, torch.nn.Identity) else _conv(blk.skip_connection, x)
    return skip + h


def _attention(blk, x, head_channels, use_proj_attn):
    b, c = x.shape[:2]
    heads = c // head_channels if head_channels else 1
    seq = blk.norm(x).reshape(b, c, -1).transpose(1, 2)  # [b, n, c]

    def split(t):
        return t.reshape(b, -1, heads, c // heads).transpose(1, 2)  # [b, heads, n, d]

    q, k, v = (split(F.linear(seq, m.weight, m.bias)) for m in (blk.to_q, blk.to_k, blk.to_v))
    probs = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(c / heads), dim=-1)
    o = (probs @ v).transpose(1, 2).reshape(b, -1, c)
    if use_proj_attn:
        o = F.linear(o, blk.proj_attn.weight, blk.proj_attn.bias)
    return o.transpose(1, 2).reshape(x.shape) + x


def unet_forward_torch(model, x: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
    """eps = DiffusionModelUNet(x, timesteps) with ATen ops and autograd (training only)."""
    ch0 = model.block_out_channels[0]
    freqs = model._freqs().to(x.device)
    ang = timesteps[:, None].float() * freqs[None, :]
    emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if ch0 % 2:
        emb = F.pad(emb, (0, 1))
    emb = model.time_embed(emb)
    h = _conv(model.conv_in, x)
    skips = [h]
    for blk, hc in zip(model.down_blocks, model.num_head_channels):
        for j, r in enumerate(blk.resnets):
            h = _resnet(r, h, emb)
            if hasattr(blk, "attentions"):
                h = _attention(blk.attentions[j], h, hc, model.use_proj_attn)
            skips.append(h)
        if blk.downsampler is not None:
            h = _conv(blk.downsampler.op, h, stride=2)
            skips.append(h)
    mid = model.middle_block
    h = _resnet(mid.resnet_1, h, emb)
    h = _attention(mid.attention, h, model.num_head_channels[-1], model.use_proj_attn)
    h = _resnet(mid.resnet_2, h, emb)
    for blk, hc in zip(model.up_blocks, reversed(model.num_head_channels)):
        for j, r in enumerate(blk.resnets):
            h = _resnet(r, torch.cat([h, skips.pop()], dim=1), emb)
            if hasattr(blk, "attentions"):
                h = _attention(blk.attentions[j], h, hc, model.use_proj_attn)
        if blk.upsampler is not None:
            h = _conv(blk.upsampler.conv, F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(model.out[2], F.silu(model.out[0](h)))


class DDPMTrainer(BaseTrainer):
    def __init__(self, args):
        super().__init__(args)
        if getattr(args, "quick_test", 0):
            print("Quick test enabled, only running on a single train and eval batch.")
        self.quick_test = bool(getattr(args, "quick_test", 0))
        self.num_epochs = args.n_epochs
        self.seed = int(args.seed)
        self.amp = bool(getattr(args, "amp", 0))
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.amp)
        # native step (hand-written HIP forward / backward / Adam) unless switched off, under AMP, or for a 3-D UNet without an MFMA tiling
        self.native = (os.environ.get("DDPM_TRAIN_NATIVE", "1") not in ("0", "off") and not self.amp
                       and native_supported(self.model) and not self.do_latent_pad)
        if self.native:
            with torch.no_grad():
                self.stepper = NativeUNetStep(self.model, lr=2.5e-5)  # base.py:156
            self._broadcast_initial_state()
            self.optimizer = self.stepper  # state_dict() / load_state_dict() in torch.optim.Adam's format
        else:
            for p in self.model.parameters():
                p.requires_grad_(True)
            self._broadcast_initial_state()
            self.optimizer = torch.optim.Adam(params=self.model.parameters(), lr=2.5e-5)  # base.py:156
        if self.found_checkpoint and self.optimizer_state:
            self.optimizer.load_state_dict(self.optimizer_state)
        self.noise_calls = 0  # stream id of the native noise generator: one per drawn tensor
        kw = dict(batch_size=args.batch_size, is_grayscale=bool(args.is_grayscale), image_size=self.image_size,
                  spatial_dimension=args.spatial_dimension, image_roi=args.image_roi)
        self.train_loader = get_data_loader(args.training_ids, rank=self.rank, world=self.world, **kw)
        self.val_loader = get_data_loader(args.validation_ids, rank=self.rank, world=self.world, **kw)
        self.gen = torch.Generator(device=self.device).manual_seed(self.seed * 7919 + self.rank)
        self.host_gen = torch.Generator().manual_seed(self.seed * 7919 + self.rank)  # native step: timesteps are drawn on the host
        self.history = []  # (epoch, mean train loss)

    def _broadcast_initial_state(self):
        """Every rank starts from rank 0's parameters and buffers (torch's default init is unseeded per process; a
        resumed run loads the same file everywhere and the broadcast is a no-op in value)."""
        if not self.ddp:
            return
        if getattr(self, "native", False) and not list(self.model.buffers()):
            self._dist_all(dist.broadcast, self.stepper.flat, src=0)  # the holders' .data are views of this buffer
            return
        tensors = [p.data for p in self.model.parameters()] + [b.data for b in self.model.buffers()]
        flat = torch.cat([t.reshape(-1).float() for t in tensors])
        dist.broadcast(flat, src=0)
        off = 0
        for t in tensors:
            t.copy_(flat[off: off + t.numel()].view_as(t).to(t.dtype))
            off += t.numel()

    # ---- one optimisation step (ddpm_trainer.py:77-101) --------------------------------------------------
    def _loss(self, images: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            images = self.vqvae_model.encode_stage_2_inputs(images).float()
            if self.do_latent_pad:
                images = F.pad(input=images, pad=self.latent_pad, mode="constant", value=0)
            b = images.shape[0]
            timesteps = torch.randint(0, self.scheduler.num_train_timesteps, (b,), device=self.device,
                                      generator=self.gen).long()
            noise = torch.randn(images.shape, device=self.device, generator=self.gen)
            noisy = self.scheduler.add_noise(original_samples=images.contiguous(), noise=noise, timesteps=timesteps,
                                             b_scale=self.b_scale)
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.amp):
            pred = unet_forward_torch(self.model, noisy, timesteps)
            # the reference regresses onto the noise whatever --prediction_type says (ddpm_trainer.py:99-100)
            return F.mse_loss(pred.float(), noise.float())

    @torch.no_grad()
    def _loss_native(self, images: torch.Tensor, backward: bool) -> torch.Tensor:
        """The same step on the native kernels: timesteps drawn on the host, noise by ddpm_randn_f32 (a pure function of seed, rank
        and draw counter), add_noise / UNet forward / MSE (/ backward) in HIP.  Returns the loss as a 1-element device tensor."""
        images = self.vqvae_model.encode_stage_2_inputs(images).float().contiguous()
        b = images.shape[0]
        timesteps = torch.randint(0, self.scheduler.num_train_timesteps, (b,), generator=self.host_gen).long()
        self.noise_calls += 1
        noise = T.randn(tuple(images.shape), self.device, self.seed * 7919 + self.rank, self.noise_calls)
        noisy = self.scheduler.add_noise(original_samples=images, noise=noise, timesteps=timesteps, b_scale=self.b_scale)
        t_dev = timesteps.to(self.device)
        if backward:
            return self.stepper.loss_and_grads(noisy, t_dev, noise)
        loss, _ = T.mse_loss_grad(self.stepper.forward(noisy, t_dev), noise)
        return loss

    def _dist_all(self, fn, flat, **kw):
        if dist.get_backend() == "gloo" and flat.is_cuda:  # test hook: two ranks on one GPU (see trainer.BaseTrainer)
            host = flat.cpu()
            fn(host, **kw)
            flat.copy_(host)
        else:
            fn(flat, **kw)  # RCCL over xGMI: one collective

    def _sync_grads(self):
        if not self.ddp:
            return
        if self.native:  # every gradient already lives in one flat buffer; the 1 / world factor rides in the Adam kernel
            self._dist_all(dist.all_reduce, self.stepper.gflat)
            return
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        if dist.get_backend() == "gloo" and flat.is_cuda:  # test hook: two ranks on one GPU (see trainer.BaseTrainer)
            host = flat.cpu()
            dist.all_reduce(host)
            flat.copy_(host)
        else:
            dist.all_reduce(flat)  # RCCL over xGMI: one collective per step
        flat /= self.world
        off = 0
        for g in grads:
            g.copy_(flat[off: off + g.numel()].view_as(g))
            off += g.numel()

    def train_epoch(self, epoch: int) -> float:
        self.model.train()
        n_local = len(self.train_loader.names)
        order = torch.randperm(n_local, generator=torch.Generator().manual_seed(self.seed + epoch))
        if self.ddp:
            # equal step counts on every rank: shards differ by at most one image, the short ones wrap around
            n_all = len(getattr(self.train_loader, "all_names", self.train_loader.names))
            n_even = -(-n_all // self.world)
            if n_local == 0:
                raise ValueError(f"rank {self.rank}: empty training shard ({n_all} images over {self.world} ranks)")
            if n_local < n_even:
                order = torch.cat([order, order[: n_even - n_local]])
        bs = self.train_loader.batch_size
        epoch_loss, epoch_step = 0.0, 0
        t0 = time.time()
        src = self.train_loader.images
        for s in range(0, len(order), bs):
            idx = order[s: s + bs]
            # a loader over images of different shapes holds a list: stack the selected items
            images = (src[idx] if torch.is_tensor(src) else torch.stack([src[int(i)] for i in idx])).to(
                self.device, non_blocking=True)
            if self.native:
                loss = self._loss_native(images, backward=True)  # every gradient has one writer: nothing to zero
                self._sync_grads()
                self.stepper.adam_step(grad_scale=1.0 / self.world if self.ddp else 1.0)
            else:
                self.optimizer.zero_grad(set_to_none=True)
                loss = self._loss(images)
                self.scaler.scale(loss).backward()  # (identity without --amp)
                self._sync_grads()
                self.scaler.step(self.optimizer)
                self.scaler.update()
            epoch_loss += loss.item()
            self.global_step += images.shape[0]
            epoch_step += images.shape[0]
            if self.quick_test:
                break
        extra = ""
        if self.native and getattr(self.stepper, "scale_adaptive", False) and self.stepper.loss_scale:
            extra = (f"; gradient scale 2^{int(math.log2(self.stepper.loss_scale))}, {self.stepper.overflow_retries} repeated backward "
                     f"passes, {self.stepper.fp32_dgrad_steps} steps on the fp32 pipe so far")
        print(f"Epoch {epoch}: loss {epoch_loss / max(epoch_step, 1):.6f} ({time.time() - t0:.1f} s{extra})")
        return epoch_loss / max(epoch_step, 1)

    @torch.no_grad()
    def val_epoch(self, epoch: int) -> float:
        self.model.eval()
        tot, n = 0.0, 0
        for batch in self.val_loader:
            img = batch["image"].to(self.device)
            tot += (self._loss_native(img, backward=False) if self.native else self._loss(img)).item()
            n += batch["image"].shape[0]
            if self.quick_test:
                break
        print(f"Validation {epoch}: loss {tot / max(n, 1):.6f}")
        return tot / max(n, 1)

    def save_checkpoint(self, path, epoch, save_message=None):
        if self.rank != 0:
            return
        checkpoint = {"epoch": epoch + 1,  # save epoch + 1, so we resume on the next epoch (base.py:170)
                      "global_step": self.global_step, "model_state_dict": self.model.state_dict(),
                      "optimizer_state_dict": self.optimizer.state_dict(), "best_loss": self.best_loss}
        print(save_message)
        torch.save(checkpoint, path)

    def train(self, args):
        self.run_dir.mkdir(parents=True, exist_ok=True)
        for epoch in range(self.start_epoch, self.num_epochs):
            epoch_loss = self.train_epoch(epoch)
            self.history.append((epoch, epoch_loss))
            if epoch_loss < self.best_loss:
                self.best_loss = epoch_loss
                self.save_checkpoint(self.run_dir / "checkpoint.pth", epoch,
                                     save_message=f"Saving checkpoint for model with loss {self.best_loss}")
            if args.checkpoint_every != 0 and (epoch + 1) % args.checkpoint_every == 0:
                self.save_checkpoint(self.run_dir / f"checkpoint_{epoch + 1}.pth", epoch,
                                     save_message=f"Saving checkpoint at epoch {epoch + 1}")
            if (epoch + 1) % args.eval_freq == 0:
                self.val_epoch(epoch)
        print("Training completed.")
        if self.ddp:
            dist.destroy_process_group()
