"""Parameter initialisation, weight-file load/save and 2D->3D inflation — mirror of reference lib/utils/net.py.

Checkpoint format (reference :252-294): a pickle `{'blobs': {unscoped_name: ndarray, ...}, 'cfg': yaml_str}`;
py2 pickles load with encoding='latin1'.  Names may carry a `_[xyz]_` branch prefix that maps onto the same
source blob (:72-92).  A 4-D (2D conv) source weight is inflated to a 5-D target by repeating along a new time
axis (:95-161): 'mean-repeat' (/kT), 'repeat', 'center-only' (all but the centre slice zero — what every shipped
config uses), 'center-only-rest-rand'.
"""
import logging
import pickle

import numpy as np
import yaml

from detectandtrack_amd.core.config import cfg

logger = logging.getLogger(__name__)


# ---- fillers (Caffe2 filler semantics used by the builders' init specs) -------------------------------------------
def _fill(kind, kw, shape, rs):
    shape = tuple(shape)
    size = int(np.prod(shape))
    if kind == 'ConstantFill':
        return np.full(shape, kw.get('value', 0.), dtype=np.float32)
    if kind == 'GaussianFill':
        return (rs.randn(*shape) * kw.get('std', 1.) + kw.get('mean', 0.)).astype(np.float32)
    if kind == 'XavierFill':       # U(-s, s), s = sqrt(3 / fan_in), fan_in = size / shape[0]
        s = np.sqrt(3.0 / (size / shape[0]))
        return rs.uniform(-s, s, shape).astype(np.float32)
    if kind == 'MSRAFill':         # N(0, sqrt(2 / fan_out)), fan_out = size / shape[1]
        return (rs.randn(*shape) * np.sqrt(2.0 / (size / shape[1]))).astype(np.float32)
    if kind == 'BilinearFill':     # detector.py:356-372
        up = kw['up_scale']
        k = 2 * up
        factor = (k + 1) // 2
        center = factor - 1 if k % 2 == 1 else factor - 0.5
        og = np.ogrid[:k, :k]
        filt = (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)
        w = np.zeros(shape, dtype=np.float32)
        w[range(shape[1]), range(shape[0]), :, :] = filt
        return w
    raise ValueError('unknown filler {}'.format(kind))


def initialize_params(model, ws, seed=None):
    """param_init_net equivalent: create every parameter of `model` in the workspace from its init spec."""
    rs = np.random.RandomState(cfg.RNG_SEED if seed is None else seed)
    for name in model.params:
        spec = model.param_specs[name]
        ws.set_param(name, _fill(spec['init'][0], spec['init'][1], spec['shape'], rs))


def synthetic_params(model, seed=3):
    """Deterministic, well-conditioned weights for parity tests / benchmarks (no checkpoints are available offline):
    ReLU-followed convs N(0, 2/fan_in), linear convs/FCs N(0, 1/fan_in), conv1 additionally /64 (inputs are
    mean-subtracted pixels), affine scale U(0.5, 1) — U(0.2, 0.4) on the last affine of a residual branch so that
    activations stay O(1..10) through the stack —, affine bias N(0, 0.1), conv/FC biases N(0, 0.05); the bilinear
    up-sampling kernel is the fixed one of detector.py:356-372."""
    rs = np.random.RandomState(seed)
    trans = cfg.RESNETS.TRANS_FUNC
    last_bn = '_branch2c_bn_s' if trans == 'bottleneck_transformation' else '_branch2b_bn_s'
    linear = ('fpn_', 'rpn_cls', 'rpn_bbox', 'cls_score', 'bbox_pred', 'kps_score')
    out = {}
    for name in model.params:
        spec = model.param_specs[name]
        shape = spec['shape']
        if spec['init'][0] == 'BilinearFill':
            out[name] = _fill('BilinearFill', spec['init'][1], shape, rs)
        elif spec.get('affine'):
            if name.endswith('_s'):
                lo, hi = (0.2, 0.4) if name.endswith(last_bn) else (0.5, 1.0)
                out[name] = rs.uniform(lo, hi, shape).astype(np.float32)
            else:
                out[name] = (rs.randn(*shape) * 0.1).astype(np.float32)
        elif len(shape) == 1:
            out[name] = (rs.randn(*shape) * 0.05).astype(np.float32)
        else:
            fan_in = int(np.prod(shape)) / shape[0]
            if name.startswith('kps_score_lowres'):   # ConvTranspose [in, out, k, k]: 4 taps reach each output
                fan_in = shape[0] * 4
            gain = 1.0 if name.startswith(linear) else 2.0
            w = rs.randn(*shape) * np.sqrt(gain / fan_in)
            if name == 'conv1_w':
                w = w / 64.0
            out[name] = w.astype(np.float32)
    return out


# ---- inflation (:95-161) ---------------------------------------------------------------------------------------------------
def inflate_weights_2d(pretrained_w, target, src_name):
    """Same-rank inflation: tile every axis that is an integer multiple and divide by the factor (:72-92)."""
    w = pretrained_w.copy()
    for d in range(target.ndim):
        if target.shape[d] % pretrained_w.shape[d] != 0:
            logger.info('Cant inflate {} ({}) for {}'.format(src_name, pretrained_w.shape, target.shape))
            return target
        t = target.shape[d] // pretrained_w.shape[d]
        reps = [1] * target.ndim
        reps[d] = t
        w = np.tile(w, reps) / t
    return w


def inflate_weights(pretrained_w, target, src_name, mode=None):
    """2D -> 3D weight inflation along a new time axis (axis -3)."""
    mode = cfg.VIDEO.WEIGHTS_INFLATE_MODE if mode is None else mode
    if target.ndim != 5:
        if target.ndim == pretrained_w.ndim:
            return inflate_weights_2d(pretrained_w, target, src_name)
        logger.info('Not trying to inflate {}'.format(src_name))
        return target
    kt = int(target.shape[-3])
    w = np.repeat(np.expand_dims(pretrained_w, axis=-3), kt, axis=-3)
    if mode == 'mean-repeat':
        w = w / float(kt)
    elif mode == 'repeat':
        pass
    elif mode == 'center-only':
        w[..., :kt // 2, :, :] = 0
        w[..., kt // 2 + 1:, :, :] = 0
    elif mode == 'center-only-rest-rand':
        sigma = 0.001
        w[..., :kt // 2, :, :] = sigma * np.random.randn(*w[..., :kt // 2, :, :].shape)
        w[..., kt // 2 + 1:, :, :] = sigma * np.random.randn(*w[..., kt // 2 + 1:, :, :].shape)
        w = w / float(kt)
    else:
        raise ValueError('Invalid INFLATE_MODE: {}'.format(mode))
    if w.shape != target.shape:
        logger.error('blob {} {} does not match weights file shape {} even after inflating'.format(
            src_name, target.shape, w.shape))
    return w


# ---- weights file ------------------------------------------------------------------------------------------------------------
def load_weights_file(path):
    with open(path, 'rb') as f:
        try:
            src = pickle.load(f)
        except UnicodeDecodeError:
            f.seek(0)
            src = pickle.load(f, encoding='latin1')
    return src['blobs'] if 'blobs' in src else src


def initialize_from_weights_file(model, ws, weights_file, momentum=None):
    """Overlay every model parameter present in the file on the ALREADY INITIALISED workspace (the reference runs
    param_init_net first, then this: :164-249).  A parameter that is absent from the file keeps its init; one whose shape
    differs is inflated onto the initialised blob and keeps its init where inflation is impossible (:95-161 return the
    workspace blob in those branches).  `<param>_momentum` blobs, when the file has them and `momentum` (a dict) is
    given, are returned through it for training.Trainer (:228-236).  One process per GPU: no broadcast needed."""
    src = load_weights_file(weights_file)
    missing = [n for n in model.params if n not in ws.params]
    if missing:   # called on a blank workspace: what the reference's param_init_net would have created
        rs = np.random.RandomState(cfg.RNG_SEED)
        for name in model.params:
            spec = model.param_specs[name]
            v = _fill(spec['init'][0], spec['init'][1], spec['shape'], rs)   # same stream as initialize_params
            if name not in ws.params:
                ws.set_param(name, v)
    kept = []
    first_init = 'trainedCOCO' in weights_file      # (:165: an ImageNet / COCO initialisation, not a checkpoint: its momentum is not restored)
    for name in model.params:
        # (:186-195) a parameter named `_[xyz]_foo` that the file does not hold under that name is initialised from the file's `foo`
        src_name = name[name.find(']_') + 2:] if (name.find(']_') >= 0 and name not in src) else name
        if src_name not in src:
            kept.append(name)
            continue
        w = np.asarray(src[src_name], dtype=np.float32)
        shape = tuple(model.param_specs[name]['shape'])
        if tuple(w.shape) != shape:
            target = np.asarray(ws.params[name], dtype=np.float32)
            w = np.asarray(inflate_weights(w, target, src_name), dtype=np.float32)
            if w is target or tuple(w.shape) != shape:
                kept.append(name)
                logger.info('%s: file shape %s cannot be inflated to %s, keeping the initialised value', name,
                            np.asarray(src[src_name]).shape, shape)
                continue
        ws.set_param(name, w)
        if momentum is not None and not first_init and src_name + '_momentum' in src:
            m = np.asarray(src[src_name + '_momentum'], dtype=np.float32)
            if tuple(m.shape) == shape:
                momentum[name] = m
    if kept:
        logger.info('%d parameters not taken from %s (kept as initialised): %s', len(kept), weights_file, ' '.join(kept))
    return kept


def save_model_to_weights_file(weights_file, model, ws, momentum=None):
    """:252-294: parameters, their `<param>_momentum` blobs (when given: name -> array) and the cfg yaml."""
    blobs = {name: np.asarray(ws.params[name]) for name in model.params}
    for name, m in (momentum or {}).items():
        blobs[name + '_momentum'] = np.asarray(m, dtype=np.float32)
    cfg_yaml = yaml.safe_dump(_plain(cfg))
    with open(weights_file, 'wb') as f:
        pickle.dump(dict(blobs=blobs, cfg=cfg_yaml), f, protocol=2)


def _plain(d):
    if isinstance(d, dict):
        return {k: _plain(v) for k, v in d.items()}
    if isinstance(d, np.ndarray):
        return d.tolist()
    if isinstance(d, tuple):
        return list(d)
    if isinstance(d, (np.floating, np.integer)):
        return d.item()
    return d
