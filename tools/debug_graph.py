#!/usr/bin/env python3
"""Debug aid: ClipGraph at the bench size, phase by phase."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa


def main():
    T, H, W = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 768, int(sys.argv[2]) if len(sys.argv) > 2 else 1344
    model, ws = bench.build('18', T, 'bf16')
    from detectandtrack_amd.core.clip_graph import ClipGraph
    clips = [bench.synthetic_clip(T, H, W, i).cuda() for i in range(2)]
    im_info = np.array([[H, W, 800.0 / 720.0]], dtype=np.float32)
    im_shape = (int(round(H / im_info[0, 2])), int(round(W / im_info[0, 2])), 3)
    st = torch.cuda.Stream()
    print('capturing', flush=True)
    g = ClipGraph(model, ws, clips[0], im_info, im_shape, stream=st)
    print('captured', flush=True)
    for i in range(4):
        g.launch(clips[i % 2])
        torch.cuda.synchronize()
        print('replayed', i, flush=True)
        r = g.results()
        print('results', i, None if r is None else len(r[0][1]), flush=True)
    from detectandtrack_amd.ops import hip_ops as ops
    with torch.cuda.stream(st):
        pr = ops.ConvProfiler(capacity=5120)
        pr.start()
    print('prof started', flush=True)
    g.launch(clips[0]); torch.cuda.synchronize(); print('replay after prof start ok', len(g.results()[0][1]), flush=True)
    import ctypes as C
    from detectandtrack_amd import libdat as L
    with torch.cuda.stream(st):
        h = ops.ctx().h
    def replay(tag):
        g.launch(clips[1]); torch.cuda.synchronize(); print('replay after %s ok' % tag, len(g.results()[0][1]), flush=True)
    tags = (C.c_int * 16)(); fl = (C.c_double * 16)(); ms = (C.c_float * 16)()
    print('prof_read', L.lib().dat_prof_read(h, 16, tags, fl, ms), flush=True)
    replay('prof_read')
    L.lib().dat_prof_enable(h, 0)
    replay('prof_enable(0)')
    mhz = C.c_double(0.0)
    L.lib().dat_prof_clock(h, C.byref(mhz))
    print('clock', mhz.value, flush=True)
    replay('prof_clock')
    # a second slot on a forked workspace, like bench's pipeline
    w1 = ws.fork()
    st1 = torch.cuda.Stream()
    g1 = ClipGraph(model, w1, clips[1], im_info, im_shape, stream=st1)
    print('captured second', flush=True)
    for i in range(3):
        g.launch(clips[0]); g1.launch(clips[1])
        print(i, len(g.results()[0][1]), len(g1.results()[0][1]), flush=True)
    print('DONE', flush=True)


if __name__ == '__main__':
    main()
