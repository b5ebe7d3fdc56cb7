"""Learning-rate schedule with warm-up (behaviour of reference lib/utils/lr_policy.py:19-114, SOLVER.* keys unchanged).

    get_lr_at_iter(it) = policy(it) * warm-up factor(it)          (float32, like the reference)

Policies (SOLVER.LR_POLICY): 'step' (BASE_LR * GAMMA^(it // STEP_SIZE)), 'steps_with_decay' (BASE_LR * GAMMA^k in the k-th
interval of STEPS), 'steps_with_lrs' (LRS[k]).  Warm-up over the first WARM_UP_ITERS iterations: 'constant' scales by
WARM_UP_FACTOR, 'linear' ramps from WARM_UP_FACTOR to 1.
"""
import bisect

import numpy as np

from detectandtrack_amd.core.config import cfg


def get_step_index(cur_iter):
    """Index k of the STEPS interval [STEPS[k], STEPS[k+1]) that holds cur_iter (the last interval ends at MAX_ITER; an iteration
    at or past MAX_ITER belongs to the last interval, like the reference's loop)."""
    s = cfg.SOLVER
    assert s.STEPS[0] == 0, 'The first step should always start at 0.'
    bounds = list(s.STEPS) + [s.MAX_ITER]
    return min(bisect.bisect_right(bounds, cur_iter), len(bounds) - 1) - 1


_POLICIES = {
    'step': lambda it: cfg.SOLVER.BASE_LR * cfg.SOLVER.GAMMA ** (it // cfg.SOLVER.STEP_SIZE),
    'steps_with_decay': lambda it: cfg.SOLVER.BASE_LR * cfg.SOLVER.GAMMA ** get_step_index(it),
    'steps_with_lrs': lambda it: cfg.SOLVER.LRS[get_step_index(it)],
}
_WARM_UP = {
    'constant': lambda frac: cfg.SOLVER.WARM_UP_FACTOR,
    'linear': lambda frac: cfg.SOLVER.WARM_UP_FACTOR * (1 - frac) + frac,
}


def get_lr_func():
    try:
        return _POLICIES[cfg.SOLVER.LR_POLICY]
    except KeyError:
        raise NotImplementedError('Unknown LR policy: {}'.format(cfg.SOLVER.LR_POLICY))


def get_lr_at_iter(it):
    lr = get_lr_func()(it)
    n_warm = cfg.SOLVER.WARM_UP_ITERS
    if it < n_warm:
        if cfg.SOLVER.WARM_UP_METHOD not in _WARM_UP:
            raise KeyError('Unknown SOLVER.WARM_UP_METHOD: {}'.format(cfg.SOLVER.WARM_UP_METHOD))
        lr *= _WARM_UP[cfg.SOLVER.WARM_UP_METHOD](it / n_warm)
    return np.float32(lr)
