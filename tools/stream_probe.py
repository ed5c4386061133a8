"""Plain streaming speeds of the box (torch elementwise kernels) to compare the apply kernel's memory skeleton with."""
import torch
n = 512 ** 3 * 3
a = torch.randn(n, dtype=torch.float64, device="cuda")
b = torch.randn(n, dtype=torch.float64, device="cuda")
c = torch.empty_like(a)
def t(fn, bytes_, name, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-22s %.3f ms  %.2f TB/s" % (name, ms, bytes_ / ms / 1e9))
t(lambda: c.copy_(a), 2 * n * 8, "copy (1R+1W)")
t(lambda: torch.add(a, b, out=c), 3 * n * 8, "add (2R+1W)")
t(lambda: c.fill_(1.0), n * 8, "fill (1W)")
t(lambda: a.sum(), n * 8, "sum (1R)")
t(lambda: torch.dot(a, b), 2 * n * 8, "dot (2R)")
