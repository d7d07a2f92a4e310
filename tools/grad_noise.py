"""Noise study of the full-config B=6 backward: error of every sampled gradient slice against the reference golden for
two consecutive runs of this build (determinism) -- run it under two builds (UNIMM_HIP_LIB) to compare realisations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_fullsize as F
from oracle.cases import grad_sample_index

gd = os.path.join(F.ROOT, "tests", "golden")
g = np.load(os.path.join(gd, "full_b6.npz")); gg = np.load(os.path.join(gd, "full_b6_grads.npz"))
model, _, _ = F.build_full(seed=5)
args, kw = F.full_b6_call(g)
params = dict(model.named_parameters())
names = [k[6:] for k in gg.files if k.startswith("grad::")]
want_names = sys.argv[1:] or [n for n in names if "c_layer.5" in n or "c_layer.1." in n]
runs = []
for rep in range(2):
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, _ = model(*args, **kw, _want_lm_scores=False)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    runs.append({n: params[n].grad.detach().double().cpu().numpy().copy() for n in want_names})
for n in want_names:
    want = gg["grad::" + n]
    def sl(a):
        a = torch.from_numpy(a)
        return (a[::4] if a.dim() == 1 else a[torch.from_numpy(grad_sample_index(tuple(a.shape))[0])][:, ::4]).numpy()
    a, b = sl(runs[0][n]), sl(runs[1][n])
    e0 = np.linalg.norm(a - want) / np.linalg.norm(want)
    rr = np.linalg.norm(runs[0][n] - runs[1][n]) / np.linalg.norm(runs[0][n])
    full = np.linalg.norm(runs[0][n])
    print(f"{n:70s} slice-err {e0:.3e}  run-to-run {rr:.1e}  |g| {full:.3e}  |slice| {np.linalg.norm(want):.3e}")
np.savez(os.environ.get("GRAD_DUMP", "/tmp/grad_dump.npz"), **{n: runs[0][n] for n in want_names})
