"""dX = gy @ W of the blocks' linear layers at 16384 tokens: the product as autograd spells it (W as stored,
[out, in]) against the same product over a transposed copy of W ([in, out], what F.linear takes), on operands
that rotate through more memory than the Infinity Cache holds."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
M = 16384
shapes = {"qkv": (4096, 1024), "o": (1024, 2048), "gate_up": (6144, 1024), "down": (1024, 3072)}  # W [out, in]
SETS = 6


def timed(fn, n=30):
    for i in range(SETS):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i % SETS)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, (n_out, n_in) in shapes.items():
    gy = [torch.randn(M, n_out, device=dev, dtype=torch.bfloat16) for _ in range(SETS)]
    w = [torch.randn(n_out, n_in, device=dev, dtype=torch.bfloat16) for _ in range(SETS)]
    wt = [x.t().contiguous() for x in w]
    x = [torch.randn(M, n_in, device=dev, dtype=torch.bfloat16) for _ in range(SETS)]
    flop = 2 * M * n_out * n_in
    t_nn = timed(lambda i: gy[i] @ w[i])
    t_tn = timed(lambda i: F.linear(gy[i], wt[i]))
    t_fwd = timed(lambda i: F.linear(x[i], w[i]))
    t_dw = timed(lambda i: gy[i].t() @ x[i])
    t_tr = timed(lambda i: w[i].t().contiguous())
    print(f"{name:8s} dX as stored {t_nn:6.1f} us ({flop / t_nn / 1e9:5.2f} PF/s) | dX over W^T {t_tn:6.1f} us ({flop / t_tn / 1e9:5.2f}) | "
          f"forward {t_fwd:6.1f} us ({flop / t_fwd / 1e9:5.2f}) | dW {t_dw:6.1f} us ({flop / t_dw / 1e9:5.2f}) | transposing W {t_tr:5.1f} us")
    del gy, w, wt, x
