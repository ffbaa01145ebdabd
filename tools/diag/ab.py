"""Diagnostic: A/B timing of alternative builds of the library on the headline loss call (interleaved rounds)."""
import ctypes, os, sys, subprocess, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from dair_pll_amd import _capi
if os.environ.get("DPLL_LIB"): _capi.LIB_PATH = os.environ["DPLL_LIB"]
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(%r, "tests", "golden", "cube_box_4096.npz"))
s = MultibodyLearnableSystem({"cube": os.path.join(%r, "assets", "cube.urdf")}, float(g["dt"]), dtype=torch.float32, device="cuda:0")
x = torch.tensor(g["x"], dtype=torch.float32, device="cuda:0"); xp = torch.tensor(g["x_plus"], dtype=torch.float32, device="cuda:0")
out = []
for rep in range(5):
    out.append(s.profile_loss_kernels(x, xp, reps=100)[0] * 1e3)
print("RESULT", sorted(out)[len(out)//2], min(out))
''' % (REPO, REPO, REPO)
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != 'default': env['DPLL_LIB'] = os.path.join(REPO, 'tools', 'diag', lib)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')]
        print(lib, line[0] if line else r.stderr[-300:])
