"""Diagnostic: which build of the loss kernel serves which batch size (lane-per-contact one-wave-per-SIMD / dense, or wide):
kernel time by HIP events for wide = 0 / 1 at a ladder of sizes.   python tools/diag/build_choice.py [cube|elbow] [f32|f64]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
w = sys.argv[1] if len(sys.argv) > 1 else 'cube'
dtype = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == 'f64') else torch.float32
g = np.load(os.path.join(REPO, 'tests', 'golden', w + '_box_4096.npz'))
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', w + '.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
x = torch.tensor(g['x'], dtype=dtype, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
for mult in (2, 4, 5, 6, 8, 12, 16):
    xb, xpb = x.repeat(mult, 1), xp.repeat(mult, 1)
    out = []
    for wide in (0, 1):
        s.set_solver(wide=wide, portfolio=1)
        ts = [s.profile_loss_kernels(xb, xpb, reps=50) for _ in range(3)]
        out.append((min(a for a, _ in ts) * 1e3, min(b for _, b in ts) * 1e3))
    print(f'{w} {dtype} {xb.shape[0]:6d} pairs: lane-per-contact {out[0][0]:.1f} + {out[0][1]:.1f} us, wide {out[1][0]:.1f} + {out[1][1]:.1f} us', flush=True)
