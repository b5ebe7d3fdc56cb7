"""Host-side cost of a training iteration (cProfile over Trainer.step; the GPU runs asynchronously, so this is the Python / ctypes /
allocator time that has to stay ahead of ~17 ms of kernels)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
from detectandtrack_amd.training import Trainer
torch.cuda.set_device(0)
model, ws = bench.build_train('18', 8, 768, 1344, 'bf16', 1, 0)
tr = Trainer(model, ws, None)
for _ in range(4):
    tr.step(1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.step(1e-4)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('10 iterations: host returned after %.1f ms/iter, GPU done after %.1f ms/iter' % (t_host * 100, t_all * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.step(1e-4)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
