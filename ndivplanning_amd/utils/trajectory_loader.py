"""Trajectory datasets with the reference's `PushDataset.__getitem__` contract
(utils/trajectory_loader.py:38-72): (images [T,3,H,W] f32 in [-1,1], states [T,25],
actions [T,4], goal [3]).

`PushDataset` reads the reference's HDF5 bundles (generate_trajectories.py:275-324) and
needs h5py + PIL, which this image lacks: it raises a clear error at construction.  The
input pipeline is a "next" row of the scope table; the training
path is exercised with `SyntheticPushDataset`, which has the same output contract, plus a
`codes` mode that yields per-frame 128-d codes instead of images (what a cache of the
frozen encoder's outputs would hold)."""
import numpy as np
import torch

from .argparse_util import listdir_nohidden


class SyntheticPushDataset(torch.utils.data.Dataset):
    """Seeded random trajectories.  mode 'images': frames ~ U[-1,1) [T,3,128,128];
    mode 'codes': frame codes ~ N(0,1) [T,128] in place of the images; mode 'frames_u8': decoded camera frames as the
    JPEG decoder leaves them, bytes [T,128,128,3] (what `PushDataset(raw_uint8=True)` yields)."""

    def __init__(self, num_trajectories, seq_length=8, mode="codes", seed=0, image_size=128):
        if mode not in ("codes", "images", "frames_u8"):
            raise ValueError("mode must be 'codes', 'images' or 'frames_u8'")
        self.n, self.seq_length, self.mode, self.seed, self.hw = int(num_trajectories), int(seq_length), mode, seed, image_size

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        gen = torch.Generator().manual_seed(self.seed * 1000003 + int(index))
        t = self.seq_length
        if self.mode == "codes":
            frames = torch.randn(t, 128, generator=gen)
        elif self.mode == "frames_u8":
            frames = torch.randint(0, 256, (t, self.hw, self.hw, 3), generator=gen, dtype=torch.uint8)
        else:
            frames = torch.rand(t, 3, self.hw, self.hw, generator=gen) * 2.0 - 1.0
        states = torch.randn(t, 25, generator=gen)
        actions = torch.rand(t, 4, generator=gen) * 2.0 - 1.0
        goal = torch.randn(3, generator=gen)
        return frames, states, actions, goal


def norm_frame(image):
    """utils/hdf5_load.py:9-11, `(ToTensor()(image) - 0.5) * 2.0`, without torchvision: ToTensor is bytes HWC -> CHW,
    float32, divided by 255."""
    t = torch.from_numpy(np.array(image, dtype=np.uint8)).permute(2, 0, 1).contiguous()
    return (t.to(torch.float32).div(255) - 0.5) * 2.0


class PushDataset(torch.utils.data.Dataset):
    """The reference's HDF5 + JPEG dataset (utils/trajectory_loader.py:17-72)."""

    def __init__(self, datadir, seq_start=0, seq_length=15, transform=None, raw_uint8=False):
        # raw_uint8: images as the decoder's bytes [T,H,W,3] instead of the normalised float tensor [T,3,H,W]; the
        # kernels apply the reference's normalisation (utils/hdf5_load.py:9-11) as they read (a quarter of the upload)
        self.raw_uint8 = bool(raw_uint8)
        try:
            import h5py  # noqa: F401
            from PIL import Image  # noqa: F401
        except ImportError as e:  # pragma: no cover - depends on the image
            raise RuntimeError("PushDataset needs h5py and PIL to read the reference's trajectory bundles "
                               "(%s); use a `synthetic:` train_data_path instead" % e)
        import h5py
        self.datadir, self.transform = datadir, transform
        self.seq_start, self.seq_length = seq_start, seq_length
        self.files = listdir_nohidden(datadir)
        counts = []
        for f in self.files:
            with h5py.File(f, "r") as h:
                counts.append(len(h))
        self.file_seq_cts = np.cumsum(counts)
        self.total_seq_ct = int(self.file_seq_cts[-1]) if counts else 0

    def __len__(self):
        return self.total_seq_ct

    def __getitem__(self, index):  # pragma: no cover - needs h5py
        import io

        import h5py
        from PIL import Image
        file_index = int(np.argmax(self.file_seq_cts > index))
        seq_index = index if file_index == 0 else index - int(self.file_seq_cts[file_index - 1])
        sl = slice(self.seq_start, self.seq_start + self.seq_length)
        with h5py.File(self.files[file_index], "r") as f:
            seq = f["trajectory_{:05d}".format(seq_index)]
            frames = []
            for b in seq["images"][sl]:
                img = Image.open(io.BytesIO(b.tobytes() if hasattr(b, "tobytes") else b))
                frames.append(torch.from_numpy(np.array(img, dtype=np.uint8)) if self.raw_uint8 else norm_frame(img))
            images = torch.stack(frames)
            states = torch.from_numpy(seq["states"][sl])
            actions = torch.from_numpy(seq["actions"][sl])
            goal = torch.from_numpy(np.array(seq["goal"]))
        if self.transform:
            images = self.transform(images)
        return images, states, actions, goal
