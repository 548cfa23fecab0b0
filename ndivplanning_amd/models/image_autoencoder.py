"""Frozen image encoder that produces the GAN's condition codes -- mirror of the reference's
`models/image_autoencoder.py::Encoder` (image_autoencoder.py:14-49): 5 x conv3x3 stride 2
(BatchNorm only after the first three; conv4_bn / conv5_bn exist as attributes but are not
applied) + a 4x4 conv to 128 channels; 3x128x128 -> 128x1x1.

It runs under `no_grad` in eval mode ahead of the GAN step (train_gan.py:75-76, 152-153) and
is outside the hand-written kernel scope this round: the convolutions
go through PyTorch-ROCm / MIOpen.  The class exists so that the reference's whole-module
encoder pickles load and so that image-mode batches can be encoded."""
import torch.nn as nn
import torch.nn.functional as F


def normal_init(m, mean, std):
    if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d)):
        m.weight.data.normal_(mean, std)
        m.bias.data.zero_()


class Encoder(nn.Module):
    _CHANNELS = (3, 64, 128, 256, 512, 1024)

    def __init__(self, d=16):
        super().__init__()
        ch = self._CHANNELS
        for i in range(5):
            setattr(self, "conv%d" % (i + 1), nn.Conv2d(ch[i], ch[i + 1], 3, 2, 1))
            setattr(self, "conv%d_bn" % (i + 1), nn.BatchNorm2d(ch[i + 1]))
        self.conv6 = nn.Conv2d(ch[5], 128, 4, 1, 0)

    def weight_init(self, mean, std):
        for name in self._modules:
            normal_init(self._modules[name], mean, std)

    def forward(self, x):
        for i in (1, 2, 3):
            x = F.relu(getattr(self, "conv%d_bn" % i)(getattr(self, "conv%d" % i)(x)))
        x = F.relu(self.conv4(x))
        x = F.relu(self.conv5(x))
        return self.conv6(x)
