"""Clip sharding across GPUs — the multi-GPU model of the reference's inference path, one process per GPU.

Reference: `multi_gpu_test_net_on_dataset` (lib/core/test_engine.py:278-308) splits the clip list into NUM_GPUS
contiguous ranges with `np.array_split` (lib/utils/subprocess.py:38), runs `tools/test_net.py --range s e` in one
subprocess per GPU and concatenates the per-range results in range order (:290-297).  There is no data-path
collective: clips are independent units.  Here the same protocol runs either as plain subprocesses (`--range`) or
under `torch.distributed` (one rank per GPU; backend "nccl" == RCCL on ROCm, "gloo" in CPU tests), where the only
collectives are the result gather and the timing barrier/MAX used by bench.py.
"""
import os

import numpy as np


def shard_range(num_items, world_size, rank):
    """[start, end) of `rank`'s contiguous range == np.array_split(range(num_items), world_size)[rank]."""
    parts = np.array_split(np.arange(num_items), world_size)
    p = parts[rank]
    if len(p) == 0:
        start = int(sum(len(q) for q in parts[:rank]))
        return start, start
    return int(p[0]), int(p[-1]) + 1


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init_process_group(backend=None):
    """Initialise torch.distributed from the launcher's env (RANK/WORLD_SIZE/MASTER_*). Returns the module or None."""
    rank, local_rank, world = env_rank_world()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if torch.cuda.is_available():
        # one rank = one GPU: bind BEFORE the process group / the workspace exist (GlobalWorkspace() allocates on
        # torch.cuda.current_device(); without this every rank would sit on GPU 0 and RCCL would see duplicate devices)
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if not dist.is_initialized():
        dist.init_process_group(backend)
    return dist


def gather_in_range_order(local_results, dist):
    """Concatenate per-rank result lists in rank (== range) order on rank 0 (test_engine.py:290-297)."""
    if dist is None:
        return list(local_results)
    world = dist.get_world_size()
    gathered = [None] * world if dist.get_rank() == 0 else None
    dist.gather_object(list(local_results), gathered, dst=0)
    if dist.get_rank() != 0:
        return None
    out = []
    for part in gathered:
        out.extend(part)
    return out


def max_over_ranks(value, dist, device=None):
    """MAX-reduce a scalar (bench.py: the slowest rank defines the step time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ('cuda' if dist.get_backend() == 'nccl' else 'cpu'))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
