"""Practical write-only ceiling: device fill of a large buffer (torch fill_, zero_) in TB/s."""
import torch

n = 48 << 30
x = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, fn in (("fill_", lambda: x.fill_(3)), ("zero_", lambda: x.zero_()),
                 ("fill_ int32 view", lambda: x.view(torch.int32).fill_(7)),
                 ("fill_ float4-ish int64 view", lambda: x.view(torch.int64).fill_(7))):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("%-28s %.3f ms  %.2f TB/s" % (name, ms, n / ms / 1e9))
