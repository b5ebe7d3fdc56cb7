"""HBM read / write rates of this box with trivial kernels (torch fill / copy / sum): the practical ceilings the HBM-bound layers are
priced against in DESIGN.md.  GPU only."""
import torch


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    for mb in (264, 1024, 4096):
        n = mb * (1 << 20) // 4
        a = torch.empty(n, dtype=torch.float32, device='cuda')
        b = torch.empty(n, dtype=torch.float32, device='cuda')
        tw = timed(lambda: a.fill_(1.0))
        tc = timed(lambda: b.copy_(a))
        tr = timed(lambda: a.sum())
        print('%5d MB: write (fill) %.2f TB/s   copy (read + write) %.2f TB/s   read (sum) %.2f TB/s'
              % (mb, n * 4 / tw / 1e12, 2 * n * 4 / tc / 1e12, n * 4 / tr / 1e12))
        del a, b


if __name__ == '__main__':
    main()
