"""GAN training script -- mirror of the reference's `train_gan.py` (same CLI, same YAML keys,
same `train(config)` entry point, same epoch log line and checkpoint files), with the loop
body (train_gan.py:115-207) replaced by `GanTrainer.step`: a few fused gfx950 kernels per
iteration, replayed from a HIP graph, with no host synchronisation inside an epoch.

What differs from the reference, on purpose:
  * the three `.cpu()` loss reads per step (train_gan.py:205-207) become one read per epoch
    of sums accumulated on the device;
  * dead work is gone: the K-fold image repeats that are never read (144-149), the back-prop
    through G inside the D step and the D weight gradients inside the G step (183, 202);
  * a ragged final batch is dropped (the reference hard-codes batch_size and crashes on it,
    train_gan.py:168-194);
  * noise comes from a counter-based generator on the device unless `training.gan.noise_source`
    is "cpu" (the reference's torch.FloatTensor(...).uniform_() stream);
  * more than one process (torch.distributed.run) trains data-parallel: `batch_size` is the
    GLOBAL batch, every rank takes batch_size / world_size trajectories of each batch
    (ndivplanning_amd/dp.py);
  * the forward-model checkpoints the reference loads and never uses (train_gan.py:79-86) are
    not loaded; visdom plotting is attempted only if visdom is importable.

Extra YAML keys (all optional): `train_data_path: synthetic:<N>[:codes|images|frames_u8]` for seeded
synthetic trajectories, `training.gan.noise_source`, `training.gan.use_graph`,
`training.gan.steps_per_launch` (iterations per HIP-graph launch, default 16; the batches of one
launch are staged into separate input slots, the arithmetic is unchanged),
`training.gan.cache_codes` (default true: encode every trajectory once, the encoder being frozen).
In data-parallel runs each rank caches only the trajectories of its own shard positions, so the
cache fills over several epochs there.
"""
import importlib
import logging
import os
from argparse import ArgumentParser

# HSA reads this when the runtime initialises (the first torch.cuda call below): dmabuf IPC is the only
# mode this platform's driver supports, and the peer-to-peer gradient exchange / RCCL need it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.utils import data  # noqa: E402

from . import dp  # noqa: E402
from .models.gan import Decoder, Discriminator  # noqa: E402
from .models.image_autoencoder import Encoder  # noqa: E402
from .trainer import GanTrainer  # noqa: E402
from .utils.argparse_util import override_dotmap  # noqa: E402
from .utils.cli_arguments.common_arguments import add_common_arguments  # noqa: E402
from .utils.file import make_paths_absolute  # noqa: E402
from .utils.trajectory_loader import PushDataset, SyntheticPushDataset  # noqa: E402


def denorm(tensor):
    return ((tensor + 1.0) / 2.0) * 255.0


def norm(image):
    return (image / 255.0 - 0.5) * 2.0


def _get(cfg, key, default):
    v = cfg.get(key, None) if hasattr(cfg, "get") else None
    return default if v is None or (isinstance(v, dict) and not v) else v


def bind_reference_class_paths():
    """Whole-module checkpoints (train_gan.py:254-266) must record the classes as `models.gan.Decoder` /
    `models.gan.Discriminator` (and the encoder as `models.image_autoencoder.Encoder`): those are the paths the
    reference's evaluation scripts unpickle (control_evaluation.py:175-176).  The root-level shims with those
    names rename the classes when imported; do that here, so that every way into train() -- the CLI entry, the
    package, a test -- writes the same pickles.  Returns the names that could not be bound."""
    missing = []
    for name, classes in (("models.gan", (Decoder, Discriminator)), ("models.image_autoencoder", (Encoder,))):
        try:
            mod = importlib.import_module(name)
        except ImportError:
            mod = None
        for cls in classes:
            if mod is not None and getattr(mod, cls.__name__, None) is cls:
                cls.__module__ = name
            else:
                missing.append("%s.%s" % (name, cls.__name__))
    if missing:
        logging.warning("checkpoints will record ndivplanning_amd.* class paths: %s do(es) not resolve to this "
                        "implementation (is the repository root, with its models/ shims, on sys.path?)", ", ".join(missing))
    return missing


def make_dataset(config):
    path = str(config.train_data_path)
    if path.startswith("synthetic:") or "/synthetic:" in path:
        spec = path[path.index("synthetic:"):].split(":")
        n = int(spec[1])
        mode = spec[2] if len(spec) > 2 else "codes"
        return SyntheticPushDataset(n, seq_length=config.trajectory_length, mode=mode, seed=int(config.random_seed))
    # decoded frames stay bytes until the first convolution reads them (ndp_encoder_forward_u8 / ndp_fm_*_u8):
    # `raw_uint8: false` restores the reference's host-side float tensors
    return PushDataset(config.train_data_path, seq_length=config.trajectory_length, raw_uint8=bool(_get(config, "raw_uint8", True)))


def load_encoder(config, device):
    """torch.load of the pretrained whole-module encoder pickle (train_gan.py:75-76).  A local,
    trusted file written by train_autoencoder.py: weights_only=False is required for module
    pickles.  Without a file (synthetic image runs) a seeded Encoder stands in."""
    path = _get(config, "image_encoder_model_path", None)
    if path and os.path.isfile(path):
        enc = torch.load(path, map_location=device, weights_only=False)
    else:
        logging.warning("no image encoder checkpoint at %s: using a seeded random Encoder", path)
        torch.manual_seed(int(config.random_seed))
        enc = Encoder()
    return enc.to(device).eval()


def encode_batch(frames, encoder, seq_length):
    """codes [B*(T-1), 256] = cat(code(current frame), code(final frame)) (train_gan.py:127-155).
    `frames` is [B,T,3,H,W] images or [B,T,128] cached codes; the target frame is encoded once
    per trajectory instead of T-1 times."""
    b, t = frames.shape[0], frames.shape[1]
    if frames.dim() == 5:
        with torch.no_grad():
            per_frame = encoder(frames.reshape((b * t,) + tuple(frames.shape[2:]))).reshape(b, t, -1)
    else:
        per_frame = frames
    cur = per_frame[:, :-1]
    tgt = per_frame[:, -1:].expand(-1, t - 1, -1)
    return torch.cat([cur, tgt], dim=2).reshape(b * (t - 1), -1).contiguous()


def epoch_batches(n_items, batch_size, generator):
    """Shuffled index batches of one epoch (ragged tail dropped): what DataLoader(shuffle=True,
    drop_last=True) does, as an explicit list so that cached epochs can follow the same order."""
    perm = torch.randperm(n_items, generator=generator)
    return [perm[i:i + batch_size] for i in range(0, n_items - batch_size + 1, batch_size)]


def train(config):
    g = config.training.gan
    random_seed = int(config.random_seed)
    num_epochs, num_sample, noise_dim = int(g.num_epochs), int(g.num_sample), int(g.noise_dim)
    batch_size, dsteps, epochs_per_stage = int(g.batch_size), int(g.discrim_steps_per_gen), int(g.epochs_per_stage)
    lr_rate = float(g.learning_rate)
    div_factor = g.get("pairwise_div_factor", None)
    if div_factor is None or isinstance(div_factor, dict):
        raise KeyError("training.gan.pairwise_div_factor is missing from the config "
                       "(the reference fails at train_gan.py:199 for such a file)")
    noise_source = _get(g, "noise_source", "device")
    use_graph = bool(_get(g, "use_graph", True))
    steps_per_launch = int(_get(g, "steps_per_launch", 16))
    cache_codes = bool(_get(g, "cache_codes", True))

    rank, world, local_rank = dp.env_world()
    if not torch.cuda.is_available():
        raise RuntimeError("train_gan needs a ROCm GPU: the HIP path has no CPU fallback")
    gpu = local_rank if world > 1 else int(_get(config, "gpu_id", 0)) % max(torch.cuda.device_count(), 1)
    device = torch.device("cuda", gpu)
    torch.cuda.set_device(device)
    dp.init_process_group(device)
    if batch_size % world != 0:
        raise ValueError("training.gan.batch_size=%d must be a multiple of the %d ranks" % (batch_size, world))
    local_batch = batch_size // world

    torch.manual_seed(random_seed)          # train_gan.py:65-66: Decoder, then Discriminator, then shuffling
    np.random.seed(random_seed)
    bind_reference_class_paths()

    display = None
    if rank == 0 and _get(config, "log_port", None):
        try:                                  # train_gan.py:68; needs a live visdom server
            from visdom import Visdom        # noqa: F401
            from .vis_tools import visualizer
            display = visualizer(port=config.log_port)
        except Exception as e:                # noqa: BLE001 - observability is optional
            logging.info("visdom plotting disabled (%s)", e)

    dataset = make_dataset(config)
    order_gen = torch.Generator().manual_seed(random_seed)
    n_batches = len(dataset) // batch_size
    if n_batches == 0:
        raise ValueError("dataset of %d trajectories is smaller than one batch of %d" % (len(dataset), batch_size))
    seq_length = int(dataset.seq_length)
    image_mode = getattr(dataset, "mode", "images") in ("images", "frames_u8")
    encoder = load_encoder(config, device) if image_mode else None

    decoder = Decoder(noise_dim=noise_dim)
    discriminator = Discriminator()
    decoder.weight_init(mean=0.0, std=0.02)            # no-op on Linear layers, as in the reference
    discriminator.weight_init(mean=0.0, std=0.02)
    decoder, discriminator = decoder.to(device), discriminator.to(device)

    flat_local = local_batch * (seq_length - 1)
    # gradient exchange of a data-parallel run: summed inside the slab-reduce kernels over hipIpc-mapped
    # peer memory when that passes its self-check on this node (the step stays one graph), RCCL
    # all-reduce between the phases otherwise
    p2p, reduce_fn, exchange = dp.make_exchange(device, world, log=logging.info)
    if world > 1 and rank == 0:
        logging.info("data parallel over %d ranks, gradient exchange: %s", world, exchange)
    trainer = GanTrainer(decoder, discriminator, flat=flat_local, num_sample=num_sample, lr=lr_rate,
                         betas=(0.5, 0.999), pairwise_div_factor=float(div_factor), discrim_steps=dsteps,
                         flat_global=flat_local * world, reduce_fn=reduce_fn, p2p=p2p,
                         use_graph=use_graph, noise_seed=random_seed * 1000 + rank,
                         steps_per_launch=steps_per_launch if (reduce_fn is None and use_graph) else 1)
    group = trainer.nslots
    # The encoder is frozen (train_gan.py:75-76, .detach() at 152-153), so a frame's code never
    # changes: with `cache_codes` every trajectory is encoded (and its actions uploaded) once, in
    # the first epoch that meets it; later epochs gather codes and actions on the device and touch
    # neither the images nor the host.
    code_cache = action_cache = cached = None
    if cache_codes:
        code_cache = torch.zeros(len(dataset), seq_length, 128, device=device)
        action_cache = torch.zeros(len(dataset), seq_length, 4, device=device)
        cached = torch.zeros(len(dataset), dtype=torch.bool)
    history = []
    # Replicas apply the same Adam update to the same summed gradients, so their parameters must stay bit-identical.
    # The in-kernel exchange has its own time-out word, but a sum that arrived WRONG (a flag overtaking its data on a
    # transport it has not met) would train on silently: compare parameter checksums across ranks after the first
    # launch and at the end of every epoch (two 16-byte all-reduces); a mismatch raises on every rank.
    lockstep = {"checked_first_launch": world == 1}

    def check_lockstep(where):
        if world > 1:
            if p2p is not None:
                p2p.check()                  # a timed-out wait is an error, not a silent wrong sum
            dp.assert_replicas_identical([trainer.g_flat, trainer.d_flat], exchange, where)

    for epoch in range(num_epochs):
        discriminator.train()
        decoder.train()
        pending = []                         # up to `group` prepared batches -> one graph launch
        batches = epoch_batches(len(dataset), batch_size, order_gen)
        lo, hi = dp.shard_bounds(batch_size, rank, world)

        def flush():
            if len(pending) == group and group > 1:
                trainer.step_many(torch.stack([p[0] for p in pending]), torch.stack([p[1] for p in pending]),
                                  None if pending[0][2] is None else torch.stack([p[2] for p in pending]))
            else:
                for c_, a_, n_ in pending:
                    trainer.step(c_, a_, n_)
            del pending[:]
            if not lockstep["checked_first_launch"]:
                lockstep["checked_first_launch"] = True
                check_lockstep("after the first launch")

        all_cached = cached is not None and bool(cached.all())
        loader = None if all_cached else iter(data.DataLoader(dataset, batch_sampler=[b.tolist() for b in batches]))
        for idx in batches:
            if all_cached:
                mine = idx[lo:hi].to(device)
                per_frame, actions = code_cache[mine], action_cache[mine]
            else:
                frames, _states, actions, _goal = next(loader)
                frames = frames[lo:hi]
                if frames.dtype != torch.uint8:                      # byte frames [B,T,128,128,3] are uploaded as they are
                    frames = frames.float()
                frames = frames.to(device, non_blocking=True)
                actions = actions[lo:hi].float().to(device, non_blocking=True)
                if frames.dim() == 5:
                    with torch.no_grad():
                        b_, t_ = frames.shape[0], frames.shape[1]
                        per_frame = encoder(frames.reshape((b_ * t_,) + tuple(frames.shape[2:]))).reshape(b_, t_, -1)
                else:
                    per_frame = frames
                if cached is not None:
                    mine = idx[lo:hi]
                    code_cache[mine.to(device)] = per_frame
                    action_cache[mine.to(device)] = actions
                    cached[mine] = True
            codes = encode_batch(per_frame, None, seq_length)
            acts = actions[:, :-1].reshape(-1, actions.size(-1))                # train_gan.py:137
            noise = None
            if noise_source == "cpu":                                           # train_gan.py:44
                # every rank has the same CPU generator state: draw the GLOBAL batch's noise, the single-process
                # stream, and keep this rank's rows (identical draws per rank would duplicate noise W-fold)
                noise = torch.FloatTensor(flat_local * world, num_sample, noise_dim).uniform_()
                noise = noise[rank * flat_local:(rank + 1) * flat_local].to(device)
            pending.append((codes, acts, noise))
            if len(pending) == group:
                flush()
        flush()
        check_lockstep("end of epoch %d" % epoch)
        sums = dp.reduce_loss_shares(trainer.pop_loss_sums(), device=device)
        d_avg, g_avg, div_avg = (v / n_batches for v in sums)                   # train_gan.py:209-211
        history.append((d_avg, g_avg, div_avg))
        if rank == 0:
            logging.info("{}, D: {:4f}, G: {:4f}, div: {:4f}".format(epoch, d_avg, g_avg, div_avg))
            if display is not None:
                display.plot("gan", "discriminator", "GAN Loss", epoch, d_avg)
                display.plot("gan", "generator", "GAN Loss", epoch, g_avg)
                display.plot("pairwise_div", "loss", "Pairwise Divergence Loss", epoch, div_avg)
            if epoch % epochs_per_stage == epochs_per_stage - 1:                # train_gan.py:249-266
                os.makedirs(config.gan_save_path, exist_ok=True)
                torch.cuda.synchronize(device)
                torch.save(discriminator, os.path.join(config.gan_save_path, "gan_discriminator_{}.pt".format(epoch)))
                torch.save(decoder, os.path.join(config.gan_save_path, "gan_decoder_{}.pt".format(epoch)))
    if p2p is not None:
        del trainer
        p2p.close()
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
