"""Network-input preparation — mirror of reference lib/utils/blob.py:40-90."""
import numpy as np

from detectandtrack_amd.core.config import cfg
import detectandtrack_amd.utils.image as image_utils


def im_list_to_blob(ims, num_frames=None):
    """List of HxWx3 (BGR, mean-subtracted) images -> NCHW blob, zero-padded to a common size (a multiple of
    FPN.COARSEST_STRIDE when FPN is on, :47-50); video models get (B, C, T, H, W) (:59-61)."""
    if not isinstance(ims, list):
        ims = [ims]
    max_shape = np.array([im.shape for im in ims]).max(axis=0)
    if cfg.FPN.FPN_ON:
        stride = float(cfg.FPN.COARSEST_STRIDE)
        max_shape[0] = int(np.ceil(max_shape[0] / stride) * stride)
        max_shape[1] = int(np.ceil(max_shape[1] / stride) * stride)
    blob = np.zeros((len(ims), max_shape[0], max_shape[1], 3), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, 0:im.shape[0], 0:im.shape[1], :] = im
    blob = blob.transpose((0, 3, 1, 2))
    if cfg.MODEL.VIDEO_ON:
        blob = image_utils.move_batch_to_time(blob, num_frames or cfg.VIDEO.NUM_FRAMES)
    return blob


def prep_im_for_blob(im, pixel_means, target_sizes, max_size):
    """Mean-subtract and scale so the short side is target_size, capped so the long side <= max_size (:71-90)."""
    im = im.astype(np.float32, copy=False)
    im = im - pixel_means
    short, long_ = np.min(im.shape[0:2]), np.max(im.shape[0:2])
    ims, scales = [], []
    for target in target_sizes:
        scale = float(target) / float(short)
        if np.round(scale * long_) > max_size:
            scale = float(max_size) / float(long_)
        # cv2.resize(im, None, None, fx=scale, fy=scale, INTER_LINEAR): size = round(n * scale), sampling step 1 / scale
        ims.append(image_utils.resize_bilinear(im, fx=scale, fy=scale))
        scales.append(scale)
    return ims, scales


def test_scale(frame_shape, target_size, max_size):
    """The scale prep_im_for_blob applies (:78-85): short side -> target_size, capped so that the long side <= max_size."""
    short, long_ = min(frame_shape[0:2]), max(frame_shape[0:2])
    scale = float(target_size) / float(short)
    if np.round(scale * long_) > max_size:
        scale = float(max_size) / float(long_)
    return scale


def frames_to_blob_on_device(frames_u8, num_frames, out=None):
    """prep_im_for_blob + im_list_to_blob for frames that are already on the GPU as a uint8 [F, h, w, 3] tensor (round 3: a clip is
    uploaded as 3 bytes per source pixel and prepared by dat_preprocess_frames, bit-identical to the host path).
    Returns (data blob fp32 [F / num_frames, 3, num_frames, H, W] -- or [F, 3, H, W] for 2D models --, scale, im_info rows)."""
    from detectandtrack_amd.ops import hip_ops as ops
    assert len(cfg.TEST.SCALES) == 1, 'single-scale inference (TTA is out of the hot-path scope)'
    h, w = int(frames_u8.shape[1]), int(frames_u8.shape[2])
    scale = test_scale((h, w), cfg.TEST.SCALES[0], cfg.TEST.MAX_SIZE)
    T = int(num_frames) if cfg.MODEL.VIDEO_ON else 1
    data, _ = ops.preprocess_frames(frames_u8, T, scale, cfg.PIXEL_MEANS, int(cfg.FPN.COARSEST_STRIDE) if cfg.FPN.FPN_ON else 0, out=out)
    n = data.shape[0]
    # (float64 rows: the `im_info` BLOB is their float32 rounding -- whoever feeds it converts --, the scale column also travels to the device
    #  glue, which divides the boxes by the double like the reference)
    im_info = np.tile(np.array([[data.shape[-2], data.shape[-1], scale]], dtype=np.float64), (n, 1))
    if not cfg.MODEL.VIDEO_ON:
        data = data.view(n, 3, data.shape[-2], data.shape[-1])
    return data, scale, im_info
