#!/usr/bin/env python3
"""Training entry point with the reference's command line (train.py of the reference).

    python train.py --model network.models.LIST --dataset datasets.Datasets.SyntheticIM2SDF -e run1
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...     # one process per GPU (RCCL)

Differences from the reference, by design: one process per GPU with DistributedDataParallel instead
of single-process nn.DataParallel (train.py:126 of the reference); TensorBoard is optional."""
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import list_amd                                          # noqa: E402
from list_amd import arguments, utils                   # noqa: E402

import torch                                             # noqa: E402
import torch.distributed as dist                         # noqa: E402

torch.manual_seed(333)


class _Module(torch.nn.Module):
    """Single-device stand-in for DataParallel/DDP: exposes `.module` like they do."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def wrap_model(model, config):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # ~16 M trainable parameters (64 MB of fp32 gradients) under warm start.  RCCL's ring
        # all-reduce over xGMI is per-link bound (~153 GB/s per link), so a few 16-MB buckets keep every link
        # busy while the HIP backward of the query path is still producing the encoders' gradients; bucket
        # views avoid the extra gradient copy.
        return torch.nn.parallel.DistributedDataParallel(
            model, device_ids=[config.device.index] if config.device.type == "cuda" else None,
            bucket_cap_mb=16, gradient_as_bucket_view=True,
            find_unused_parameters=True)       # vox_encoder.bn.2 is never used in forward
    return _Module(model)


def train_epoch(epoch, executor, optimizer, data_iter, config, writer=None):
    totals = {"total_loss": 0.0}
    t_epoch = time.time()
    steps = 0
    # Losses are the reference's FULL-BATCH values on every rank (executors.LIST.calc_loss gathers the SDF shards:
    # one RCCL all-gather per step), so the epoch mean that picks best_model_train is the same number everywhere.
    for batch_idx, batch in enumerate(data_iter):
        t_iter = time.time()
        _, batch_loss = executor.train(batch=batch, calc_loss=True)
        loss = sum(v for k, v in batch_loss.items() if "ignore" not in k)
        for k, v in batch_loss.items():
            totals[k] = totals.get(k, 0.0) + float(v.detach())
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        totals["total_loss"] += float(loss.detach())
        steps += 1
        if (batch_idx + 1) % config.plot_every_batch == 0 or batch_idx == len(data_iter) - 1:
            now = time.time()
            eta = (now - t_epoch) / (batch_idx + 1) * len(data_iter) - (now - t_epoch)
            parts = ", ".join(f"{k}: {float(v.detach()):9.5f}" for k, v in batch_loss.items())
            print(f"Epoch: {epoch + 1:03d}||{config.epochs}, batch: {batch_idx + 1:03d}||{len(data_iter)}, "
                  f"{parts}, batch_total_loss: {float(loss.detach()):9.5f} batch_time: {now - t_iter:0.5f} "
                  f"ETA: {int(eta // 60):02d}m:{int(eta % 60):02d}s")
        if config.max_steps and steps >= config.max_steps:
            break
    mean = totals["total_loss"] / max(steps, 1)
    print(f"{config.exp_name} Train: Epoch {epoch + 1:03d}||{config.epochs}, loss: {mean:9.5f} "
          f"epoch_time: {time.time() - t_epoch:0.5f}")
    if writer is not None:
        for k, v in totals.items():
            writer.add_scalar(f"Train: Mean {k}", v / max(steps, 1), epoch)
    return mean


def _summary_writer(path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(path)
    except Exception:
        return None


def _warm_start(net, config):
    """Load the coarse predictor's image encoder / point decoder and freeze them (reference train.py:175-228)."""
    base = "./results/coarse_prediciton_Pix3D/checkpoints/" if "Pix3D" in config.exp_name \
        else "./results/coarse_prediciton/checkpoints/"
    ime, pd = base + "best_IME_test.pt.tar", base + "best_PD_test.pt.tar"
    if os.path.exists(ime) and os.path.exists(pd):
        sd = torch.load(ime, map_location="cpu")["state_dict"]
        net.im_encoder.load_state_dict(sd)
        net.im_encoder2.load_state_dict(sd)
        net.point_decoder.load_state_dict(torch.load(pd, map_location="cpu")["state_dict"])
        print(config.exp_name, "warm-start checkpoints loaded")
    else:
        print("warm start requested but", ime, "not found: freezing randomly initialised encoders")
    for p in list(net.im_encoder.parameters()) + list(net.point_decoder.parameters()):
        p.requires_grad = False


def train(config):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = config.cuda and torch.cuda.is_available()
    if world > 1:
        dist.init_process_group("nccl" if use_cuda else "gloo")
    if use_cuda:
        torch.cuda.set_device(local_rank if world > 1 else config.gpu)
        config.device = torch.device("cuda", torch.cuda.current_device())
    else:
        config.device = torch.device("cpu")
        config.cuda = False

    model = utils.get_class(config.model)(config).to(config.device)
    # Order of the reference (train.py:141-228): the optimizer covers ALL parameters (a reference-written
    # best_model_train.pt.tar holds Adam state for every one of them; frozen parameters simply never get a
    # gradient and Adam skips them); a resume checkpoint wins over the warm start.  The warm start -- with its
    # freezing of im_encoder / point_decoder -- sits INSIDE `if config.load_pretrain:` there (train.py:152,177-228) and
    # only runs when there is no best_model_train.pt.tar; it then writes one at epoch -1 (train.py:216-218), so a
    # restart before the first best-save RESUMES that file and trains the encoders unfrozen.  Same here, quirk included.
    resume = config.checkpoint_dir + "best_model_train.pt.tar"
    resuming = bool(config.load_pretrain and os.path.exists(resume))
    warm = bool(config.load_pretrain and config.warm_start and not resuming)
    if warm:
        _warm_start(model, config)
    model = wrap_model(model, config)

    trainset = utils.get_class(config.dataset)(config, "train")
    sampler = torch.utils.data.distributed.DistributedSampler(trainset) if world > 1 else None
    train_iter = torch.utils.data.DataLoader(trainset, batch_size=config.train_batch_size,
                                             shuffle=sampler is None, sampler=sampler,
                                             num_workers=config.num_workers, drop_last=True)
    optimizer = torch.optim.Adam(model.parameters(), lr=config.lr,
                                 betas=(config.beta1, 0.999), weight_decay=config.weight_decay)
    epoch, best_train = 0, 1e3
    if warm and ((not dist.is_initialized()) or dist.get_rank() == 0):
        utils.save_checkpoint(-1, model, optimizer, best_train, resume)      # reference train.py:216-218
        print("Initial checkpoint saved.")
    if resuming:
        epoch, model, optimizer, best = utils.load_checkpoint(resume, model, optimizer)
        if best is not None:
            best_train = float(best)          # (the reference forgets this and overwrites its best model after a resume)
        print(f"pretrained model loaded at epoch: {epoch}, best train loss: {best}")
    rank0 = (not dist.is_initialized()) or dist.get_rank() == 0
    writer = _summary_writer(config.results_dir + "/summary") if rank0 else None

    executor = utils.get_class(config.model.replace("model", "executor"))(config, model)
    while epoch < config.epochs:
        if not config.skip_train:
            executor.model.train()
            if sampler is not None:
                sampler.set_epoch(epoch)
            loss = train_epoch(epoch, executor, optimizer, train_iter, config, writer)
            if rank0:
                if (epoch + 1) % config.save_every_epoch == 0:
                    utils.save_checkpoint(epoch, executor.model, optimizer, loss,
                                          config.checkpoint_dir + f"/model_{epoch + 1}.pt.tar")
                if best_train > loss:
                    best_train = loss
                    utils.save_checkpoint(epoch, executor.model, optimizer, loss, resume)
        epoch += 1
        if config.max_steps:
            break
    if dist.is_initialized():
        dist.destroy_process_group()
    return best_train


if __name__ == "__main__":
    cfg = arguments.get_args()
    utils.ensure_dir(cfg.checkpoint_dir)
    with open(utils.ensure_dir(cfg.results_dir + "code/") + "command.txt", "a+") as fp:
        fp.write(time.strftime("%m/%d/%Y %H:%M:%S") + " --> " + " ".join(sys.argv) + "\n")
    train(cfg)
