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
