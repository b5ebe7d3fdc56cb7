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


def momentum_correction(cur_lr, new_lr):
    """The factor the update history V is scaled by when the learning rate changes from cur_lr to new_lr, or None
    (reference lib/modeling/detector.py:606-616 _SetNewLr / :618-640 _CorrectMomentum): MomentumSGDUpdate keeps V = mu V + lr grad, so V
    carries the lr it was built with; on a change by more than SOLVER.SCALE_MOMENTUM_THRESHOLD (either way) the reference multiplies V
    by new_lr / cur_lr -- unless SOLVER.SCALE_MOMENTUM is off or the old lr is (about) zero, which covers the first iteration (its `lr`
    blob starts at 0, model_builder.py:959)."""
    if cur_lr is None or cur_lr == new_lr or not cfg.SOLVER.SCALE_MOMENTUM or not cur_lr > 1e-7:
        return None
    ratio = float('inf') if new_lr == 0 else max(new_lr / cur_lr, cur_lr / new_lr)
    if ratio > cfg.SOLVER.SCALE_MOMENTUM_THRESHOLD:
        return new_lr / cur_lr
    return None
