"""Experiment: the encoder's 14x14, 3-channel first convolution as-is versus with the 14 horizontal taps
folded into channels (42 input channels, 14x1 kernel).  Prints forward and weight-gradient times."""
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
B, S, K, Cout = 64, 224, 14, 32
dt = torch.bfloat16


def timeit(fn, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


x = torch.rand([B, S, S, 3], device=dev).to(dt)
w = (torch.randn([K, K, 3, Cout], device=dev) * 0.05)
pt = (K - 1) // 2
xp = F.pad(x.permute(0, 3, 1, 2), (pt, K - 1 - pt, pt, K - 1 - pt))  # [B,3,S+13,S+13] channels_last view
xp = xp.contiguous(memory_format=torch.channels_last)
wa = w.to(dt).permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def run_a():
    return F.conv2d(xp, wa)


ya = run_a()
g = torch.randn_like(ya)
print("a fwd  %.3f ms" % timeit(run_a))
print("a wrw  %.3f ms" % timeit(lambda: torch.autograd.grad(run_a(), wa, g)) )

# fold the horizontal taps: X2[b, y, x, kx*3 + c] = xp[b, y, x + kx, c]
xn = xp.permute(0, 2, 3, 1)  # [B, S+13, S+13, 3] contiguous NHWC


def fold():
    v = xn.as_strided([B, S + K - 1, S, K * 3], [xn.stride(0), xn.stride(1), xn.stride(2), 1])
    return v.contiguous().permute(0, 3, 1, 2)  # NCHW-shaped, channels_last memory


x2 = fold()
print("fold   %.3f ms" % timeit(fold), tuple(x2.shape), x2.is_contiguous(memory_format=torch.channels_last))
# w2[o, kx*3 + c, ky, 0] = w[ky, kx, c, o]
wb = w.to(dt).permute(3, 1, 2, 0).reshape(Cout, K * 3, K, 1).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def run_b():
    return F.conv2d(x2, wb)


yb = run_b()
print("max diff", float((ya.float() - yb.float()).abs().max()), float(ya.float().abs().max()))
print("b fwd  %.3f ms" % timeit(run_b))
print("b wrw  %.3f ms" % timeit(lambda: torch.autograd.grad(run_b(), wb, g)))

# pad the folded channels to 48
x3 = F.pad(x2.permute(0, 2, 3, 1), (0, 6)).permute(0, 3, 1, 2)
wc = F.pad(wb.detach().permute(0, 2, 3, 1), (0, 6)).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last).requires_grad_(True)
print(x3.shape, x3.is_contiguous(memory_format=torch.channels_last))
run_c = lambda: F.conv2d(x3, wc)
print("c fwd  %.3f ms" % timeit(run_c))
print("c wrw  %.3f ms" % timeit(lambda: torch.autograd.grad(run_c(), wc, g)))
