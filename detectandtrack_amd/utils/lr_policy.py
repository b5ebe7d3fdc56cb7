"""Learning-rate schedule with warm-up — mirror of reference lib/utils/lr_policy.py:19-114."""
import numpy as np

from detectandtrack_amd.core.config import cfg


def get_lr_at_iter(it):
    lr = get_lr_func()(it)
    if it < cfg.SOLVER.WARM_UP_ITERS:
        method = cfg.SOLVER.WARM_UP_METHOD
        if method == 'constant':
            warmup_factor = cfg.SOLVER.WARM_UP_FACTOR
        elif method == 'linear':
            alpha = it / cfg.SOLVER.WARM_UP_ITERS
            warmup_factor = cfg.SOLVER.WARM_UP_FACTOR * (1 - alpha) + alpha
        else:
            raise KeyError('Unknown SOLVER.WARM_UP_METHOD: {}'.format(method))
        lr *= warmup_factor
    return np.float32(lr)


def lr_func_steps_with_lrs(cur_iter):
    return cfg.SOLVER.LRS[get_step_index(cur_iter)]


def lr_func_steps_with_decay(cur_iter):
    return cfg.SOLVER.BASE_LR * cfg.SOLVER.GAMMA ** get_step_index(cur_iter)


def lr_func_step(cur_iter):
    return cfg.SOLVER.BASE_LR * cfg.SOLVER.GAMMA ** (cur_iter // cfg.SOLVER.STEP_SIZE)


def get_step_index(cur_iter):
    assert cfg.SOLVER.STEPS[0] == 0, 'The first step should always start at 0.'
    steps = cfg.SOLVER.STEPS + [cfg.SOLVER.MAX_ITER]
    for ind, step in enumerate(steps):
        if cur_iter < step:
            break
    return ind - 1


def get_lr_func():
    name = 'lr_func_' + cfg.SOLVER.LR_POLICY
    if name not in globals():
        raise NotImplementedError('Unknown LR policy: {}'.format(cfg.SOLVER.LR_POLICY))
    return globals()[name]
