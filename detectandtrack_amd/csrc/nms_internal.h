// Library-internal NMS entry (defined in proposals.hip) shared with detections.hip; not part of the C ABI.
#pragma once
#include "dat_common.h"

// bytes of device scratch dat_nms_impl needs for up to `cap` boxes of T frames
__attribute__((visibility("hidden"))) size_t dat_nms_ws_bytes(int cap, int T);
// Greedy NMS of dets [n, 4T+1] (any order unless presorted): n on the host, or -- when n_dev != nullptr -- *n_dev <= cap read on the
// device (no host round trip).  ws: dat_nms_ws_bytes(cap, T) bytes, or nullptr for the context workspace.  keep: int32[cap],
// num_keep: int32[1] (device).  strict / presorted: the `_nms` convention (IoU > thr, rows visited as given).
__attribute__((visibility("hidden"))) int dat_nms_impl(dat_ctx* ctx, hipStream_t st, char* ws, const float* dets, int n, const int* n_dev,
                                                       int cap, int T, float thresh, int strict, int presorted, int* keep, int* num_keep);
// The same over n_images independent box sets in one launch sequence (one block row per image): image i reads dets + i * dets_stride
// floats and n_dev[i * n_dev_stride], uses ws + i * ws_stride (ws_stride >= dat_nms_ws_bytes; ws nullptr: the context workspace) and
// writes keep + i * keep_stride / num_keep[i * num_stride].
__attribute__((visibility("hidden"))) int dat_nms_impl_batch(dat_ctx* ctx, hipStream_t st, char* ws, size_t ws_stride, const float* dets,
                                                             size_t dets_stride, int n, const int* n_dev, int n_dev_stride, int cap,
                                                             int T, float thresh, int strict, int presorted, int* keep,
                                                             int keep_stride, int* num_keep, int num_stride, int n_images);
