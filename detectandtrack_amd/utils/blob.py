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


class FrameBlob(object):
    """The `data` blob of a forward that has NOT been materialised (round 6, cfg.HIP.STEM_FROM_UINT8): the uploaded uint8 frames [F, h, w, 3]
    plus the geometry prep_im_for_blob / im_list_to_blob would give them.  The fused stem reads the frames directly
    (dat_stem_conv_pool_u8 evaluates the pre-processing arithmetic in its patch loader, bit-identical); anything else that wants the blob
    -- an unfused stem, FetchBlob('data') -- calls materialise() (dat_preprocess_frames).  `shape` is the blob's."""

    def __init__(self, frames_u8, T, scale, out_hw, pad_hw, five_d, pixel_means):
        self.frames, self.T, self.scale = frames_u8, int(T), float(scale)
        self.out_hw, self.pad_hw, self.five_d = tuple(out_hw), tuple(pad_hw), bool(five_d)
        self.pixel_means = np.asarray(pixel_means, dtype=np.float64)
        F = int(frames_u8.shape[0])
        self.N = F // self.T
        self.shape = (self.N, 3, self.T) + self.pad_hw if self.five_d else (F, 3) + self.pad_hw
        self.device = frames_u8.device

    def data_ptr(self):
        return self.frames.data_ptr()

    def materialise(self):
        """The fp32 blob dat_preprocess_frames writes for these frames: [N, 3, T, H, W], or [F, 3, H, W] for 2D models."""
        import torch
        out = torch.empty((self.N, 3, self.T) + self.pad_hw, dtype=torch.float32, device=self.frames.device)
        data = _preprocess_into(self.frames, self.T, self.scale, self.pixel_means, self.out_hw, self.pad_hw, out)
        return data if self.five_d else data.view((self.N * self.T, 3) + self.pad_hw)


def _preprocess_into(frames, T, scale, pixel_means, out_hw, pad_hw, out):
    import ctypes as C
    from detectandtrack_amd.ops import hip_ops as ops
    F, h, w, _ = [int(v) for v in frames.shape]
    means = (C.c_double * 3)(*[float(v) for v in np.asarray(pixel_means, dtype=np.float64).reshape(-1)[:3]])
    ops.ctx().call('dat_preprocess_frames', ops._stream(), ops._ptr(frames), F, int(T), h, w, C.c_double(scale), C.c_double(scale),
                   int(out_hw[0]), int(out_hw[1]), int(pad_hw[0]), int(pad_hw[1]), means, ops._ptr(out))
    return out


def frames_to_blob_on_device(frames_u8, num_frames, out=None, lazy=False):
    """prep_im_for_blob + im_list_to_blob for frames that are already on the GPU as a uint8 [F, h, w, 3] tensor (round 3: a clip is
    uploaded as 3 bytes per source pixel and prepared by dat_preprocess_frames, bit-identical to the host path).
    Returns (data blob fp32 [F / num_frames, 3, num_frames, H, W] -- or [F, 3, H, W] for 2D models --, scale, im_info rows)."""
    from detectandtrack_amd.ops import hip_ops as ops
    assert len(cfg.TEST.SCALES) == 1, 'single-scale inference (TTA is out of the hot-path scope)'
    h, w = int(frames_u8.shape[1]), int(frames_u8.shape[2])
    scale = test_scale((h, w), cfg.TEST.SCALES[0], cfg.TEST.MAX_SIZE)
    T = int(num_frames) if cfg.MODEL.VIDEO_ON else 1
    pad = int(cfg.FPN.COARSEST_STRIDE) if cfg.FPN.FPN_ON else 0
    if lazy and cfg.HIP.get('STEM_FROM_UINT8', True):
        # the blob is described, not written: the fused stem reads the uint8 frames (FrameBlob)
        oh, ow = int(np.rint(h * scale)), int(np.rint(w * scale))
        ph, pw = (oh, ow) if not pad else (int(np.ceil(oh / float(pad)) * pad), int(np.ceil(ow / float(pad)) * pad))
        fb = FrameBlob(frames_u8, T, scale, (oh, ow), (ph, pw), cfg.MODEL.VIDEO_ON, cfg.PIXEL_MEANS)
        im_info = np.tile(np.array([[ph, pw, scale]], dtype=np.float64), (fb.N if cfg.MODEL.VIDEO_ON else int(frames_u8.shape[0]), 1))
        return fb, scale, im_info
    data, _ = ops.preprocess_frames(frames_u8, T, scale, cfg.PIXEL_MEANS, pad, out=out)
    n = data.shape[0]
    # (float64 rows: the `im_info` BLOB is their float32 rounding -- whoever feeds it converts --, the scale column also travels to the device
    #  glue, which divides the boxes by the double like the reference)
    im_info = np.tile(np.array([[data.shape[-2], data.shape[-1], scale]], dtype=np.float64), (n, 1))
    if not cfg.MODEL.VIDEO_ON:
        data = data.view(n, 3, data.shape[-2], data.shape[-1])
    return data, scale, im_info
