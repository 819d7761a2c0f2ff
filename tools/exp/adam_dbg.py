import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flatland_marl_amd._lib import call, ptr, query
for Rp, T1, r4 in ((70, 151, 20), (100, 151, 24)):
    n = query("flm_dqn_param_count", 3 * Rp, T1, 2)
    g0 = torch.randn(n, device="cuda")
    rcb = torch.rand(64, device="cuda")
    K1 = 64 * 3 * Rp
    a1_w = 64 * T1 + 64; a1_b = a1_w + 512 * K1; v1_w = a1_b + 512; v1_b = v1_w + 512 * K1
    full = g0.clone()
    for w, b in ((a1_w, a1_b), (v1_w, v1_b)):
        blk = full[w:w + 512 * K1].view(512, 64, 3, Rp)
        blk[:, :, :, r4:] = (rcb.view(1, 64, 1, 1) * full[b:b + 512].view(512, 1, 1, 1)).expand(512, 64, 3, Rp - r4)
    part = full.clone()
    for w in (a1_w, v1_w):
        part[w:w + 512 * K1].view(512, 64, 3, Rp)[:, :, :, r4:] = float("nan")
    outs = []
    for fused in (0, 1):
        p = torch.randn(n, generator=torch.Generator().manual_seed(1)).cuda()
        t = torch.randn(n, generator=torch.Generator().manual_seed(2)).cuda()
        m = torch.randn(n, generator=torch.Generator().manual_seed(3)).cuda() * 1e-3
        v = torch.rand(n, generator=torch.Generator().manual_seed(4)).cuda() * 1e-6
        if fused:
            call("flm_dqn_adam_soft_update_rails", ptr(p), ptr(part), ptr(m), ptr(v), ptr(t), n, 1e-3, 0.9, 0.999, 1e-8, 3, 0.01, Rp, T1, 2, r4, ptr(rcb), None, None, None, None, 0)
        else:
            call("flm_dqn_adam_soft_update", ptr(p), ptr(full), ptr(m), ptr(v), ptr(t), n, 1e-3, 0.9, 0.999, 1e-8, 3, 0.01, 0)
        torch.cuda.synchronize()
        outs.append((p, t, m, v))
    for name, x, y in zip("ptmv", *outs):
        bad = (x != y).nonzero().flatten()
        print(Rp, name, "mismatches", bad.numel(), "of", n)
        if bad.numel():
            for i in bad[:6].tolist() + bad[-3:].tolist():
                j = i - a1_w if i < a1_b else i - v1_w
                print("  i", i, "i%4", i % 4, "o", j // K1, "c", (j % K1) // (3 * Rp), "h", j % (3 * Rp), x[i].item(), y[i].item())
