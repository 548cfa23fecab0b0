"""Diagnostic (GPU box): in-kernel phase stamps of k_g_fwd / k_d from a -DNDP_STAMPS build
of the library (cdna_hip_programming.md section 7, In-kernel stamps).  Prints, per phase, the
median over workgroups of the elapsed wall time (100 MHz clock) and shader cycles."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ndivplanning_amd import _build, _capi

lib_path = os.path.join(_build.LIB_DIR, "libndp_hip_stamps.so")
extra = os.environ.get("NDP_CXXFLAGS", "").split()
cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DNDP_STAMPS"] + extra + [
       "-Wno-unused-value", "-Wno-pass-failed", os.path.join(_build.CSRC, "ndp_kernels.hip"), "-o", lib_path]
subprocess.check_call(cmd)
_build.LIB_PATH = lib_path
lib = _capi.load()
raw = ctypes.CDLL(lib_path)

from ndivplanning_amd.models.gan import Decoder, Discriminator
from ndivplanning_amd.trainer import GanTrainer
from oracle import gan_oracle as O
dev = "cuda:0"
batch, k = int(os.environ.get("B", 64)), int(os.environ.get("K", 6))
g, d = O.init_params(0, 2)
dec, dis = Decoder(2), Discriminator(); dec.load_state_dict(g); dis.load_state_dict(d)
dec, dis = dec.to(dev), dis.to(dev)
codes, actions, noise = O.synthetic_batch(0, batch, k, steps=1)
tr = GanTrainer(dec, dis, flat=codes.shape[0], num_sample=k, use_graph=False)
tr.codes.copy_(codes); tr.actions.copy_(actions); tr.noise.copy_(noise[0])
for _ in range(20):
    tr.step()
torch.cuda.synchronize()
stamps = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
KID = {"g": 1, "ga": 1, "d": 2, "da": 2, "db": 2, "w": 4, "wa": 4, "wb": 4, "pa": 5, "pb": 6}
assert raw.ndp_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()), KID.get(os.environ.get("WHICH", "g"), 0)) == 0

def report(name, fn, nphase, nwg, first=0, skip=()):
    for _ in range(int(os.environ.get("WARM", "0"))):
        fn()
    stamps.zero_(); torch.cuda.synchronize()
    fn(); torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 32, 2)[first:first + nwg]
    clk, wall = s[:, :, 0].astype(np.float64), s[:, :, 1].astype(np.float64)
    t0 = wall[:, 0].min()
    print("%s: %d workgroups; first start -> last end %.2f us; start skew %.2f us" % (
        name, nwg, (wall[:, nphase].max() - t0) / 100.0, (wall[:, 0].max() - t0) / 100.0))
    for i in range(nphase):
        if i in skip or i + 1 in skip:
            continue
        dw = (wall[:, i + 1] - wall[:, i]) / 100.0
        dc = clk[:, i + 1] - clk[:, i]
        print("   phase %2d: median %.2f us (max %.2f)  %8.0f shader cycles  => %.2f GHz" % (
            i, np.median(dw), dw.max(), np.median(dc), np.median(dc) / max(np.median(dw), 1e-9) / 1e3))

m = codes.shape[0] * k
mpad = (m + 31) // 32 * 32
rt = 2 if m > 16384 else 1
which = os.environ.get("WHICH", "g")
if which == "w":
    # D weight gradient: 19 jobs x chunks; run the module backward (1 pass over m rows)
    a = torch.repeat_interleave(actions, k, dim=0).to(dev)
    c = torch.repeat_interleave(codes, k, dim=0).to(dev)
    def fb():
        dis.zero_grad()
        dis(a, c).sum().backward()
    nch = min((512 + 18) // 19, mpad // 64, 64)
    report("k_wgrad[D, 1 pass] [0 setup,1 main loop,2 lds write,3 reduce+store]", fb, 4, 19 * nch)
elif which == "wa":
    # D weight gradient inside the real phase A (2 passes): k_wgrad is the last stamping kernel
    nch = min(512 // 19, 2 * mpad // 64, 64)
    report("k_wgrad[D, 2 pass] [0 setup,1 main loop,2 lds write,3 reduce+store]", lambda: tr._phase_a(True), 4, 19 * nch)
    # per job: workgroup b -> xcd = b & 7, idx = b >> 3, job = idx % 19, chunk = xcd + 8 * (idx // 19)
    nchunks = int(os.environ.get("CH", 24))
    nblk = 8 * ((nchunks + 7) // 8) * 19
    w = stamps.cpu().numpy().reshape(-1, 32, 2)[:nblk][:, :, 1].astype(np.float64)
    t0 = w[w[:, 0] > 0, 0].min()
    names = ["fc1c0", "fc1c1", "fc1c2", "fc1c3", "fc1act", "fc2a", "fc2b"] + ["fc3_%d" % i for i in range(8)] + ["fc4_%d" % i for i in range(4)]
    for job in range(19):
        rows = [b for b in range(nblk) if (b >> 3) % 19 == job and w[b, 0] > 0]
        st, en = w[rows, 0] - t0, w[rows, 4] - t0
        main = w[rows, 2] - w[rows, 1]
        print("   job %2d %-7s: start %.2f..%.2f us, main loop median %.2f max %.2f, end median %.2f max %.2f"
              % (job, names[job], st.min() / 100, st.max() / 100, np.median(main) / 100, main.max() / 100,
                 np.median(en) / 100, en.max() / 100))
elif which == "wb":
    nch = min(512 // 26, mpad // 64, 64)
    tr._phase_a(True)
    report("k_wgrad[G] [0 setup,1 main loop,2 lds write,3 reduce+store]", lambda: tr._phase_b(), 4, 26 * nch)
elif which == "pa":
    report("k_phase_a role 0 [0 load,1 G fc1..fc4+fc5,2 a_hat->XT,3 D fc1,4 D fc2+fc3,5 D fc4+loss,6 stores,7 dg4+dg3,8 dg2+stores]",
           lambda: tr._phase_a(True), 8, mpad // 16)
    nt = mpad // 16
    w0 = stamps.cpu().numpy().reshape(-1, 32, 2)[:nt][:, :, 1].astype(np.float64)
    print("role 0, inside phase 2->3: a_hat->XT + sync %.2f us | dw2.preload issue %.2f | fc1 k-loop+epilogue (wave 0) %.2f | barrier %.2f"
          % tuple(np.median(w0[:, b] - w0[:, a_]) / 100 for a_, b in ((2, 9), (9, 10), (10, 11), (11, 3))))
    print("role 0, G forward: fc1 %.2f us | fc2 %.2f | fc3 %.2f | fc4 %.2f | store h4 + fc5 + actions %.2f"
          % tuple(np.median(w0[:, b] - w0[:, a_]) / 100 for a_, b in ((1, 12), (12, 13), (13, 14), (14, 15), (15, 2))))
    print("role 0, G fc3 stage: fc4 ring issue %.2f us | store h2 %.2f | k-loop + epilogue %.2f | barrier %.2f"
          % tuple(np.median(w0[:, b] - w0[:, a_]) / 100 for a_, b in ((13, 16), (16, 17), (17, 18), (18, 14))))
    print("role 0, G fc4 stage: D fc1 ring issue %.2f us | store h3 %.2f | k-loop + epilogue %.2f | barrier %.2f"
          % tuple(np.median(w0[:, b] - w0[:, a_]) / 100 for a_, b in ((14, 19), (19, 20), (20, 21), (21, 15))))
    print("role 0, prologue: fc1 ring issue %.2f us | code tile (load, LDS write) %.2f | noise %.2f | barrier %.2f"
          % tuple(np.median(w0[:, b] - w0[:, a_]) / 100 for a_, b in ((0, 24), (24, 25), (25, 26), (26, 1))))
    print("role 0, tail: store h4 %.2f us | fc5 (VALU) %.2f | barrier + rest %.2f"
          % tuple(np.median(w0[:, b] - w0[:, a_]) / 100 for a_, b in ((15, 22), (22, 23), (23, 2))))
    nr = (codes.shape[0] + 15) // 16                       # role 1: one workgroup per tile of the FLAT distinct real rows
    if nr > 0:
        both = stamps.cpu().numpy().reshape(-1, 32, 2)[:nt + nr][:, :, 1].astype(np.float64)
        t0 = both[:, 0].min()
        wall = both[nt:]
        print("role 1 (%d tiles of distinct real rows): start median %.2f us after first start, end median %.2f (max %.2f); D fc1 done at %.2f, fwd done %.2f"
              % (nr, np.median(wall[:, 0] - t0) / 100, np.median(wall[:, 8] - t0) / 100, (wall[:, 8] - t0).max() / 100,
                 np.median(wall[:, 3] - t0) / 100, np.median(wall[:, 4] - t0) / 100))
        print("role 1 stages: prologue (ring issue, code tile, actions, barrier) %.2f us | D fc1 %.2f | barrier %.2f | fc2+fc3 %.2f | fc4+loss %.2f | stores %.2f | narrow+dg3 %.2f | dg2+stores %.2f"
              % tuple(np.median(wall[:, b] - wall[:, a_]) / 100 for a_, b in ((0, 10), (10, 11), (11, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8))))
        r0 = both[:nt]
        print("role 0: end median %.2f us (max %.2f)" % (np.median(r0[:, 8] - t0) / 100, (r0[:, 8] - t0).max() / 100))
elif which == "pb":
    tr._phase_a(True)
    report("k_phase_b [0 load,1 D' fwd fc1-3,2 fc4+loss,3 D' dgrad,4 dA,5 G acts load,6 dY4 narrow,7 gg4,8 gg3,9 gg2,10 stores]",
           lambda: tr._phase_b(), 11, mpad // 16)
elif which == "ga":
    report("k_g_fwd in phase A (packed weights) [0 load,1 fc1,2 fc2,3 fc3,4 fc4,5 fc5,6 store]",
           lambda: tr._phase_a(True), 7, mpad // (16 * rt))
elif which == "da":
    report("k_d<1,2> in phase A (packed) [0 load,1 fc1,2 fc2,3 fc3,4 fc4,5 loss,6 store,7 dg4,8 dg3,9 dg2,10 store]",
           lambda: tr._phase_a(True), 11, mpad // 16)
elif which == "db":
    tr._phase_a(True)
    report("k_d<1,1> in phase B (packed) [0 load,1 fc1,2 fc2,3 fc3,4 fc4,5 loss,6 store,7 dg4,8 dg3,9 dg2,10 store]",
           lambda: tr._phase_b(), 11, mpad // (16 * rt))
elif which == "g":
    # phase A launches k_g_fwd first; the k_d launch that follows overwrites the stamp buffer,
    # so call the forward alone through the module API
    z = torch.cat([torch.repeat_interleave(codes, k, dim=0), noise[0].reshape(m, -1)], dim=1).to(dev)
    report("k_g_fwd (module path, native weight layout)  [0 load,1 fc1,2 fc2,3 fc3,4 fc4,5 fc5,6 store]", lambda: dec(z), 7, mpad // (16 * rt))
else:
    a = torch.repeat_interleave(actions, k, dim=0).to(dev).requires_grad_(True)
    c = torch.repeat_interleave(codes, k, dim=0).to(dev)
    def fb():
        dis(a, c).sum().backward(inputs=[a])
    report("k_d fwd+bwd [0 load,1 fc1,2 fc2,3 fc3,4 fc4,5 loss,6 store,7 dg4,8 dg3,9 dg2,10 store]",
           fb, 11, mpad // (16 * rt))
