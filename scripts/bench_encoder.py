"""Time the image encoder on the GPU box: hand-written kernels vs PyTorch-ROCm / MIOpen operators, per-kernel split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ndivplanning_amd import _capi
from ndivplanning_amd.models.image_autoencoder import Encoder
from oracle import encoder_oracle as EO

dev = "cuda:0"
n = int(os.environ.get("N", 1024))
enc = Encoder()
enc.load_state_dict(EO.init_encoder_state(1, bn_seed=2), strict=False)
enc = enc.to(dev).eval()
x = (torch.rand(n, 3, 128, 128, device=dev) * 2 - 1)
flop = 2.0 * 311.2e6 * n

def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

with torch.no_grad():
    t_hip = timed(lambda: enc(x))
    skip_lib = os.environ.get("SKIP_LIB") == "1"          # (under rocprofv3: keeps MIOpen's find pass out of the trace)
    t_lib = float("nan") if skip_lib else timed(lambda: enc._forward_torch(x))
    if not skip_lib:
        a, b = enc(x), enc._forward_torch(x)
        print("max |hip - miopen| / max|ref| = %.2e" % float((a - b).abs().max() / b.abs().max()))
    print("n=%d  hip %.3f ms = %.1f TFLOP/s (%.1f %% of 157.3)   miopen %.3f ms = %.1f TFLOP/s" % (
        n, t_hip * 1e3, flop / t_hip / 1e12, 100 * flop / t_hip / 157.3e12, t_lib * 1e3, flop / t_lib / 1e12))
    _capi.timing_enable(True)
    enc(x)
    torch.cuda.synchronize()
    macs = {"k_enc_conv1": 7.08e6, "k_conv_gemm[2]": 75.5e6, "k_conv_gemm[3]": 75.5e6, "k_conv_gemm[4]": 75.5e6,
            "k_conv_gemm[5]": 75.5e6, "k_conv_gemm[6]": 2.1e6}
    for name, (ms, cnt) in _capi.timing_collect().items():
        tf = 2 * macs[name] * n / (ms * 1e-3) / 1e12 if name in macs else 0
        print("   %-18s %8.3f ms (%d launches)  %6.1f TFLOP/s" % (name, ms, cnt, tf))
    _capi.timing_enable(False)
