"""Forward-model training script -- mirror of the reference's `train_forward_model.py` (same CLI, same YAML keys
`training.forward.*`, same `train(config)` entry point, epoch log line and checkpoint files), with the loop body
(train_forward_model.py:98-112) replaced by `ForwardModelTrainer.step`: forward, MSE, backward and Adam as gfx950
kernels (csrc/ndp_forward_model.inc).

What differs from the reference, on purpose:
  * the per-step `loss.cpu()` read (train_forward_model.py:113-114) becomes one read per epoch of a sum kept on the
    device;
  * the StepLR scheduler the reference constructs and never steps (train_forward_model.py:86) is not constructed;
  * visdom image panels are attempted only if visdom is importable;
  * `train_data_path: synthetic:<N>` gives seeded synthetic trajectories (the HDF5 loader needs h5py, see
    utils/trajectory_loader.py);
  * more than one process (torch.distributed.run) trains data-parallel: `batch_size` is the GLOBAL batch, every rank
    takes batch_size / world_size trajectories of each (identically shuffled) batch, the flat gradient is averaged over
    RCCL between backward and Adam (ndivplanning_amd/dp.py::mean_all_reduce).  BatchNorm statistics are per rank, as with
    torch's DistributedDataParallel; a final batch that does not split evenly is dropped; rank 0 saves.
The whole module is saved every `epochs_per_stage` epochs as the reference does (train_forward_model.py:151-163), after
the trainer's flat vectors are written back into it."""
import importlib
import logging
import os
from argparse import ArgumentParser

import numpy as np
import torch
from torch.utils import data

from . import dp
from .forward_trainer import ForwardModelTrainer
from .models.forward_encoder import Decoder, Encoder, ForwardAutoencoder
from .train_gan import _get, denorm, make_dataset, norm  # noqa: F401
from .utils.argparse_util import override_dotmap
from .utils.cli_arguments.common_arguments import add_common_arguments
from .utils.file import make_paths_absolute


def bind_reference_class_paths():
    """Whole-module checkpoints must record `models.forward_encoder.ForwardAutoencoder` (/ Encoder / Decoder): the
    path the reference's evaluation scripts unpickle (control_evaluation.py:177-180).  See train_gan.py."""
    missing = []
    try:
        mod = importlib.import_module("models.forward_encoder")
    except ImportError:
        mod = None
    for cls in (ForwardAutoencoder, Encoder, Decoder):
        if mod is not None and getattr(mod, cls.__name__, None) is cls:
            cls.__module__ = "models.forward_encoder"
        else:
            missing.append("models.forward_encoder.%s" % cls.__name__)
    if missing:
        logging.warning("checkpoints will record ndivplanning_amd.* class paths: %s do(es) not resolve to this "
                        "implementation (is the repository root, with its models/ shims, on sys.path?)", ", ".join(missing))
    return missing


def _require(config, dotted):
    """The value at `a.b.c`, or a KeyError that names the key: a missing key reads as an empty map (DotMap's behaviour,
    which the reference relies on elsewhere), and float() / int() of that is a TypeError about the wrong thing."""
    node = config
    for part in dotted.split("."):
        node = node.get(part, None) if isinstance(node, dict) else None
        if node is None:
            break
    if node is None or (isinstance(node, dict) and not node):
        raise KeyError("%s is missing from the config (the reference's config/default.yaml has it)" % dotted)
    return node


def train(config):
    for key in ("num_epochs", "learning_rate", "batch_size", "epochs_per_stage"):
        _require(config, "training.forward." + key)
    for key in ("random_seed", "train_data_path", "forward_save_path"):
        _require(config, key)
    f = config.training.forward
    random_seed = int(config.random_seed)
    lr_rate, num_epochs, batch_size = float(f.learning_rate), int(f.num_epochs), int(f.batch_size)
    epochs_per_stage = int(f.epochs_per_stage)
    rank, world, local_rank = dp.env_world()
    if not torch.cuda.is_available():
        from . import _capi
        raise _capi.NdpError("train_forward_model needs a ROCm GPU (gpu_id %r); there is no CPU path" % (config.gpu_id,))
    gpu = local_rank if world > 1 else config.gpu_id
    if world > 1 and os.environ.get("NDP_BENCH_ONE_GPU") == "1":      # rehearsal: all ranks on one card
        gpu = 0
    device = torch.device("cuda", gpu) if isinstance(gpu, int) else torch.device(gpu)
    torch.cuda.set_device(device)
    dp.init_process_group(device)
    if batch_size % world != 0:
        raise ValueError("training.forward.batch_size=%d must be a multiple of the %d ranks" % (batch_size, world))
    local_batch = batch_size // world
    bind_reference_class_paths()
    torch.manual_seed(random_seed)                                   # train_forward_model.py:60-61
    np.random.seed(random_seed)
    display = None
    try:
        from .vis_tools import visualizer                            # optional (visdom)
        display = visualizer(port=config.log_port)
    except Exception:                                                # pragma: no cover - depends on the image
        display = None

    dataset = make_dataset(config)
    if getattr(dataset, "mode", "images") not in ("images", "frames_u8"):
        raise ValueError("the forward model trains on images: use `synthetic:<N>:images` or an HDF5 directory")
    loader = data.DataLoader(dataset, batch_size=batch_size, shuffle=True)

    model = ForwardAutoencoder().to(device)                          # train_forward_model.py:67-70
    model.decoder.weight_init(mean=0.0, std=0.02)
    model.encoder.weight_init(mean=0.0, std=0.02)
    model.train()
    if world > 1:                                                    # one set of initial weights: rank 0's
        import torch.distributed as dist
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    # data parallel: the mean of the ranks' gradients, either per gradient bucket on a communication stream while the
    # backward pass is still running (default) or as ONE collective between backward and Adam
    # (`training.forward.grad_exchange: single`); two ranks give bit-identical parameters either way
    exchange = f.get("grad_exchange", None) if hasattr(f, "get") else None
    exchange = "bucketed" if exchange is None or (isinstance(exchange, dict) and not exchange) else str(exchange)
    if exchange not in ("bucketed", "single"):
        raise ValueError("training.forward.grad_exchange must be 'bucketed' or 'single', got %r" % (exchange,))
    # `batch_size` is the GLOBAL batch: BatchNorm normalises over all ranks' images (exact integer all-reduce of the
    # statistics accumulators), so W ranks train the single-process step; `training.forward.sync_batchnorm: false` keeps
    # the statistics per rank (torch DistributedDataParallel without SyncBatchNorm: 20 fewer small collectives per step)
    sync_bn = f.get("sync_batchnorm", None) if hasattr(f, "get") else None
    sync_bn = True if sync_bn is None or (isinstance(sync_bn, dict) and not sync_bn) else bool(sync_bn)
    trainer = ForwardModelTrainer(model, batch=local_batch, lr=lr_rate, betas=(0.5, 0.999),
                                  sync_batchnorm_world=world if (world > 1 and sync_bn) else 1,
                                  reduce_fn=dp.mean_all_reduce(world) if world > 1 and exchange == "single" else None,
                                  bucket_reduce=dp.BucketedMeanAllReduce(world) if world > 1 and exchange == "bucketed" else None)

    history = []
    step = 0
    for epoch in range(num_epochs):
        trainer.loss_sum.zero_()
        pairs = 0
        for images, _, actions, _ in loader:
            if world > 1:
                if images.shape[0] != batch_size:                    # ragged final batch: does not split evenly
                    continue
                lo, hi = dp.shard_bounds(batch_size, rank, world)
                images, actions = images[lo:hi], actions[lo:hi]
            images = images.to(device, non_blocking=True)
            if images.dtype != torch.uint8:                          # byte frames [B,T,128,128,3]: normalised by the kernels
                images = images.float()
            actions = actions.to(device, non_blocking=True).float()
            for image_num in range(dataset.seq_length - 1):          # train_forward_model.py:98-112
                trainer.step(images[:, image_num].contiguous(), images[:, image_num + 1].contiguous(),
                             actions[:, image_num].contiguous())
                step += 1
                pairs += 1
        avg_loss = float(trainer.loss_sum.item()) / max(pairs, 1)    # (seq_length - 1) * len(loader) terms
        if world > 1:
            avg_loss = dp.reduce_loss_shares([avg_loss / world], device=device)[0]
        history.append(avg_loss)
        if display is not None:                                      # pragma: no cover
            display.plot("loss", "train", "Forward Model Loss", epoch, avg_loss)
        logging.info("{}, {}: reconstruction loss per epoch: {}".format(epoch, step, avg_loss))
        if epoch % epochs_per_stage == epochs_per_stage - 1 and rank == 0:   # train_forward_model.py:151-163
            os.makedirs(config.forward_save_path, exist_ok=True)
            trainer.sync_to_module()
            torch.cuda.synchronize(device)
            torch.save(model, os.path.join(config.forward_save_path, "forward_autoencoder_{}.pt".format(str(epoch))))
    trainer.sync_to_module()
    trainer.close()
    train.last_trainer = trainer                                       # tests: the replica's flat vectors
    return history


def main(argv=None):
    parser = ArgumentParser(description="Interact with your training script")
    parser = add_common_arguments(parser)
    args = parser.parse_args(argv)
    config = override_dotmap(args, "config_file")
    config = make_paths_absolute(os.getcwd(), config, log_not_exist=True)
    return train(config)


if __name__ == "__main__":
    main()
