"""Clip assembly for video models -- the rule of reference lib/utils/video.py:149-201 (get_clip).

The reference scores a video with ONE clip per key frame: the clip of key frame k holds the frames at offsets
floor(-(T-1)/2) ... floor((T-1)/2) around it, VIDEO.TIME_INTERVAL apart -- for even T one more frame before the key frame than after it, the
key frame always at index T // 2 -- and a position whose frame does not exist (before the first / after the last frame of the video) takes
the nearest clip position towards the key frame that does (:183-196): border frames are replicated.  The dataset layer that builds tube
ground truth around this (:66-147) is out of scope."""
import math


def clip_frame_ids(key_frame, first_frame, last_frame, num_frames, time_interval=1):
    """The `num_frames` frame numbers of the clip around `key_frame` in a video whose frames are numbered first_frame .. last_frame
    (every frame present)."""
    assert first_frame <= key_frame <= last_frame
    half = (num_frames - 1) / 2.0
    offsets = list(range(int(math.floor(-half)), int(math.floor(half)) + 1))
    assert len(offsets) == num_frames and offsets[num_frames // 2] == 0
    ids = []
    for dt in offsets:
        sgn = 1 if dt > 0 else -1
        for d in range(abs(dt), -1, -1):        # the position itself, else the next one towards the key frame, ...
            f = key_frame + sgn * d * time_interval
            if first_frame <= f <= last_frame:
                break
        ids.append(f)
    return ids
