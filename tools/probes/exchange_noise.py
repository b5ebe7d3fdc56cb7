"""Diagnostic: bf16 training, the GRADIENTS of one iteration -- how far are two IDENTICAL runs apart (float-atomic noise of the direct weight-gradient
kernels), and how far is the run with the overlapped one-rank exchange (per-bucket deferred finish) from them?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', DAT_FORCE_EXCHANGE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
import torch.distributed as dist  # noqa: E402
from tests.test_gpu_train import _ddp_model, _ddp_clip, _ddp_feed  # noqa: E402
from detectandtrack_amd.training import Trainer  # noqa: E402

Trainer.BUCKET_BYTES = 16 << 20
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'


def run(with_dist):
    model, ws = _ddp_model(1, dtype=dtype)
    _ddp_feed(ws, _ddp_clip(0))
    tr = Trainer(model, ws, dist if with_dist else None)
    tr.step(0.0)                                   # lr 0: one forward + backward (+ exchange); the gradients stay in the arena
    torch.cuda.synchronize()
    return tr, {n: tr.arena[n].clone() for n in tr.trainable}, {n: torch.zeros_like(tr.arena[n]) for n in tr.trainable}


def cmp(a, b, init, tag):
    worst = []
    for n in a:
        step = float((b[n] - init[n]).abs().max())
        d = float((a[n] - b[n]).abs().max())
        if step > 0:
            worst.append((d / step, n, d, step))
    worst.sort(reverse=True)
    print(tag, ['%s %.3f (d %.2e step %.2e)' % (n, r, d, s) for r, n, d, s in worst[:4]])


_, r1, init = run(False)
_, r2, _ = run(False)
dist.init_process_group('nccl', rank=0, world_size=1)
_, r3, _ = run(True)
dist.destroy_process_group()
cmp(r1, r2, init, 'plain vs plain   :')
cmp(r3, r1, init, 'exchange vs plain:')
cmp(r3, r2, init, 'exchange vs plain2:')
