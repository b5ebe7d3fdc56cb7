"""Clip sources for the training input pipeline — the image half of reference lib/roi_data/minibatch.py:59-103.

A source is what `roi_data.loader.RoIDataLoader` calls from its worker threads: `source(i) -> (data, entry, im_scale)` with
`data` the mean-subtracted, scaled, stride-padded fp32 clip (1, 3, T, H, W).  `RoidbClipSource` serves a roidb whose entries
carry their frames as arrays (`entry['image']`: list of T HxWx3 uint8 BGR frames — decoding files is the dataset layer, out
of scope here, core/test_engine.load_clip) plus the usual ground truth (`boxes`, `gt_classes`, `gt_keypoints`, ...).
"""
import numpy as np

from detectandtrack_amd.core.config import cfg
import detectandtrack_amd.utils.blob as blob_utils


class RoidbClipSource(object):
    def __init__(self, roidb, seed=None):
        self.roidb = roidb
        self._seed = cfg.RNG_SEED if seed is None else seed

    def __len__(self):
        return len(self.roidb)

    @property
    def widths(self):
        return [e['width'] for e in self.roidb]

    @property
    def heights(self):
        return [e['height'] for e in self.roidb]

    def __call__(self, i):
        entry = self.roidb[i]
        frames = entry['image'] if isinstance(entry['image'], (list, tuple)) else [entry['image']]
        assert all(isinstance(f, np.ndarray) for f in frames), 'roidb frames must be decoded arrays (no image reader offline)'
        # one scale per clip, drawn like minibatch.py:66-67 but from a per-clip stream (workers run out of order)
        scales = cfg.TRAIN.SCALES
        target = scales[np.random.RandomState((self._seed + 7 * i) % (2 ** 32)).randint(0, len(scales))]
        per_frame, im_scale = [], None
        for f in frames:
            if entry.get('flipped', False):
                f = f[:, ::-1, :]
            ims, sc = blob_utils.prep_im_for_blob(f, cfg.PIXEL_MEANS, [target], cfg.TRAIN.MAX_SIZE)
            assert im_scale is None or im_scale == sc[0], 'frames of one clip must share their size'
            per_frame.append(ims[0])
            im_scale = sc[0]
        data = blob_utils.im_list_to_blob(per_frame, num_frames=len(per_frame) if cfg.MODEL.VIDEO_ON else None)
        return np.ascontiguousarray(data, dtype=np.float32), entry, float(im_scale)
