"""Train a `small` 32x32x1 DDPM on synthetic blobs with the product training loop (row f-3), then run the HIP
reconstruction path + scorer on it: the AUROC of a TRAINED model against two OOD sets (development / evidence tool;
log kept under profiles/).

    python tools/train_synthetic.py [--epochs 60] [--n_train 2048] [--out /tmp/ddpm_synth]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pandas as pd  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--n_train", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--out", default="/tmp/ddpm_synth")
    ap.add_argument("--skip", type=int, default=16)
    a = ap.parse_args()
    import train_ddpm
    import reconstruct as rcli
    from ddpm_ood_amd import ood
    from ddpm_ood_amd.train import DDPMTrainer
    from ddpm_ood_amd.trainer import Reconstruct

    sched = ["--beta_schedule", "scaled_linear_beta", "--beta_start", "0.0015", "--beta_end", "0.0195"]
    common = ["--output_dir", a.out, "--model_name", "fashionmnist_synthetic_trained", "--is_grayscale", "1"]
    targs = train_ddpm.parse_args(common + sched + [
        "--training_ids", f"synthetic:blobs:n={a.n_train}:seed=1", "--validation_ids", "synthetic:blobs:n=64:seed=10",
        "--n_epochs", str(a.epochs), "--batch_size", str(a.batch), "--eval_freq", "20", "--checkpoint_every", "0"])
    t0 = time.time()
    tr = DDPMTrainer(targs)
    tr.train(targs)
    print(f"trained {a.epochs} epochs in {time.time() - t0:.1f} s; loss {tr.history[0][1]:.5f} -> {tr.history[-1][1]:.5f}")
    del tr
    torch.cuda.empty_cache()
    rargs = rcli.parse_args(common + sched + [
        "--validation_ids", "synthetic:blobs:n=64:seed=10", "--in_ids", "synthetic:blobs:n=64:seed=11",
        "--out_ids", "synthetic:noise:n=64:seed=12:name=MNIST,synthetic:speckle:n=64:seed=13:mix=10:name=FashionMNIST_vflip,"
                     "synthetic:speckle:n=64:seed=14:mix=3:name=FashionMNIST_hflip",
        "--inference_skip_factor", str(a.skip), "--batch_size", "64"])
    rec = Reconstruct(rargs)
    rec.reconstruct(rargs)
    aucs = ood.main(argparse.Namespace(output_dir=a.out, model_name=rargs.model_name, max_t=1000, min_t=0))
    print("AUROC (trained model): noise = %.4f, speckle 10%% = %.4f, speckle 3%% = %.4f" % (
        aucs["MNIST"], aucs["FashionMNIST_vflip"], aucs["FashionMNIST_hflip"]))
    df = pd.read_csv(Path(a.out) / rargs.model_name / "ood" / "results_in.csv")
    print("mean in-distribution MSE per t:", df.groupby("t")["mse"].mean().round(5).to_dict())


if __name__ == "__main__":
    main()
