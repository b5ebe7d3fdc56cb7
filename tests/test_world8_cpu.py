"""World-8 readiness on the CPU (gloo, eight real processes): the two places where the path meets torch.distributed, at the rank count of
BASELINE's 8-GPU configurations -- (1) embarrassingly-parallel inference: contiguous clip ranges per rank
(/root/reference lib/utils/subprocess.py:38-63 np.array_split), range-order merge on rank 0 (lib/core/test_engine.py:286-297), with clip
counts that do not divide by 8, fewer clips than ranks, and none; (2) the training exchange (lib/modeling/model_builder.py:931-942): the
bucketed, overlapped gradient all-reduce with >= 6 buckets against the serial sum.  RCCL itself needs a multi-GPU node
(tests/test_gpu_train.py::test_two_ranks_nccl arms itself there)."""
import os

import numpy as np

WORLD = 8
CLIP_COUNTS = (0, 1, 7, 9, 4001)


def _shard_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from detectandtrack_amd.utils import dist as du
    from detectandtrack_amd.core.test_engine import merge_range_results
    dist = du.init_process_group('gloo')
    out = {}
    for n in CLIP_COUNTS:
        s, e = du.shard_range(n, world, rank)
        # what test_net returns for the range [s, e): per class a list with one entry per clip; clip i is marked by its index
        local = {'all_boxes': [[], [np.full((1 + i % 3, 5), i, np.float32) for i in range(s, e)]],
                 'all_keyps': [[], [[np.full((4, 17), i, np.float32)] * (1 + i % 3) for i in range(s, e)]], 'cfg': 'yaml'}
        parts = du.gather_in_range_order([local], dist)
        t = du.max_over_ranks(float(e - s), dist)                  # the busiest rank's clip count
        if rank == 0:
            merged = merge_range_results(parts)
            out[n] = ([int(b[0, 0]) for b in merged['all_boxes'][1]], [b.shape[0] for b in merged['all_boxes'][1]],
                      [int(k[0][0, 0]) for k in merged['all_keyps'][1]], len(merged['all_boxes'][0]), merged['cfg'], t,
                      [len(p['all_boxes'][1]) for p in parts])
        else:
            assert parts is None
        dist.barrier()
    if rank == 0:
        q.put(out)
    dist.destroy_process_group()


def _run(target, world, port_salt, collect_from_all):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + port_salt) % 1000)
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world if collect_from_all else 1)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_inference_sharding_and_range_order_merge_gloo_world8():
    """0 / 1 / 7 / 9 / 4001 clips over 8 ranks: every rank's range is np.array_split's, ranks beyond the clip count are EMPTY and still
    take part in the gather, rank 0's merge holds every clip exactly once in dataset order, and the MAX-reduced load is ceil(n / 8)."""
    (out,) = _run(_shard_worker, WORLD, 401, False)
    for n in CLIP_COUNTS:
        order, rows, kp_order, n_bg, cfg_kept, tmax, per_rank = out[n]
        want = [len(p) for p in np.array_split(np.arange(n), WORLD)]
        assert per_rank == want and sum(per_rank) == n, (n, per_rank)
        assert order == list(range(n)) == kp_order and rows == [1 + i % 3 for i in range(n)]
        assert n_bg == 0 and cfg_kept == 'yaml'
        assert tmax == float(-(-n // WORLD))
        if n < WORLD:
            assert per_rank.count(0) == WORLD - n                 # empty ranks


N_ELEM = 40000
BUCKETS = [(0, 9000), (9000, 9000), (9000, 17000), (17000, 17003), (17003, 26000), (26000, 33000), (33000, 39999), (39999, 40000)]


def _exchange_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from detectandtrack_amd.training import GradExchange
    g = torch.Generator().manual_seed(900 + rank)
    final = torch.randn(N_ELEM, generator=g)                       # what this rank's backward will have produced at the end
    results = {}
    for mode in ('serial', 'overlap'):
        flat = torch.full((N_ELEM,), float('nan'))                 # a gradient is garbage until its producer has run
        x = GradExchange(flat, BUCKETS, dist, overlap=(mode == 'overlap'))
        x.begin()
        for k, (lo, hi) in enumerate(BUCKETS):                     # the backward pass: bucket k becomes final, is handed over, the pass goes on
            flat[lo:hi] = final[lo:hi]
            x.ready(k)
            if mode == 'overlap' and k + 1 < len(BUCKETS):
                nlo, nhi = BUCKETS[k + 1]
                assert nhi == nlo or bool(torch.isnan(flat[nlo:nhi]).all())      # (nothing touched a bucket that is not final yet)
        x.finish()
        results[mode] = (flat.clone().numpy(), list(x.order))
    q.put((rank, final.numpy(), results))
    dist.destroy_process_group()


def test_overlapped_bucket_exchange_equals_the_serial_sum_gloo_world8():
    """GradExchange with 8 ranks and 8 buckets (one empty, one of a single element, one of three): buckets start in completion order while
    later ones are still being written; every rank ends with the sum over the ranks of every element -- equal to the serial exchange bit for
    bit and on every rank, and to the float64 sum of the eight contributions within fp32 rounding."""
    res = sorted(_run(_exchange_worker, WORLD, 577, True), key=lambda t: t[0])
    assert [r[0] for r in res] == list(range(WORLD))
    exact = np.sum(np.stack([r[1].astype(np.float64) for r in res]), axis=0)
    ref = res[0][2]['serial'][0]
    np.testing.assert_allclose(ref, exact, rtol=0, atol=4e-6)
    for _, _, r in res:
        for mode in ('serial', 'overlap'):
            got, order = r[mode]
            assert not np.isnan(got).any()
            np.testing.assert_array_equal(got, ref)                # (the same reduction on every rank, in both modes)
            assert order == list(range(len(BUCKETS)))
