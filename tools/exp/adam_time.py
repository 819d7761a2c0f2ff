"""Times of the optimizer pass in its three forms (every moment; gradient under the empty rows formed in the pass; compact
form with packed moments and gradient) beside a plain copy and a memset of the same buffers (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from flatland_marl_amd._lib import call, ptr, query
Rp, T1, r4 = 70, 151, 20
n = query("flm_dqn_param_count", 3 * Rp, T1, 2)
p, g, m, v, t = (torch.randn(n, device="cuda") for _ in range(5))
v.abs_()
rcb = torch.rand(64, device="cuda")
pa = [torch.rand(65536, 2, device="cuda") for _ in range(2)]
def timed(f, reps=30):
    for _ in range(5): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in ev:
        x.record(); f(); y.record()
    torch.cuda.synchronize()
    return float(np.median([x.elapsed_time(y) for x, y in ev])) * 1e3
print("plain  %.1f us" % timed(lambda: call("flm_dqn_adam_soft_update", ptr(p), ptr(g), ptr(m), ptr(v), ptr(t), n, 1e-3, 0.9, 0.999, 1e-8, 3, 0.01, 0)))
print("rails  %.1f us" % timed(lambda: call("flm_dqn_adam_soft_update_rails", ptr(p), ptr(g), ptr(m), ptr(v), ptr(t), n, 1e-3, 0.9, 0.999, 1e-8, 3, 0.01, Rp, T1, 2, r4, ptr(rcb), None, None, None, None, 0)))
mpk, vpk = (torch.rand(2 * 512 * 64 * 3 * r4, device="cuda") for _ in range(2))
print("compact %.1f us" % timed(lambda: call("flm_dqn_adam_soft_update_rails", ptr(p), ptr(g), ptr(m), ptr(v), ptr(t), n, 1e-3, 0.9, 0.999, 1e-8, 3, 0.01, Rp, T1, 2, r4, ptr(rcb), ptr(pa[0]), ptr(pa[1]), ptr(mpk), ptr(vpk), 0)))
print("copy p->t (2x55MB) %.1f us" % timed(lambda: t.copy_(p)))
print("memset %.1f us" % timed(lambda: t.zero_()))
