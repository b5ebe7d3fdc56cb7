#!/usr/bin/env python3
"""Build the reference's own Cython NMS / IoU and its AffineChannelNd CUDA op into oracle/_ref/ (test infrastructure).

Recipe (run here, in the container that has /root/reference):
  * read  /root/reference/lib/utils/cython_nms.pyx and cython_bbox.pyx WHERE THEY LIE;
  * the only edit is the two NumPy aliases that NumPy 2 removed (`np.int`,
    `np.int_t` -> `np.int64`, `np.int64_t`; reference cython_nms.pyx:45,48-49) —
    applied in memory, the patched text goes ONLY to oracle/_ref/ (git-ignored);
  * cythonize + compile with gcc (no -march flags, no fp contraction) into
    oracle/_ref/ref_cython_nms*.so and ref_cython_bbox*.so.

  * (round 5) /root/reference/lib/ops/affine_channel_nd_op.cu -- the one floating-point operator whose source IS in the
    reference tree -- is compiled WHERE IT LIES with hipcc for gfx950 against the Caffe2 stand-in of oracle/ref_affine/shim
    (Tensor / Operator / CUDA_1D_KERNEL_LOOP as Caffe2's public headers define them), together with the C entry points of
    oracle/ref_affine/ref_affine_driver.hip, into oracle/_ref/libref_affine.so: the reference's own kernels AND its own
    RunOnDevice() bodies run on the MI355X (tests/test_gpu_kernels.py).

Nothing is copied into tracked files.  On the GPU box /root/reference does not
exist; the prebuilt .so files travel with the snapshot and this script is a no-op.
"""
import os
import re
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
REF = '/root/reference/lib/utils'
SRCS = {'ref_cython_nms': 'cython_nms.pyx', 'ref_cython_bbox': 'cython_bbox.pyx'}


def _patch(text):
    text = re.sub(r'\bnp\.int_t\b', 'np.int64_t', text)
    text = re.sub(r'\bnp\.int\b(?!\d|_)', 'np.int64', text)
    return text


def build(force=False):
    if not os.path.isdir(REF):
        return False  # GPU box: use prebuilt artefacts
    os.makedirs(OUT, exist_ok=True)
    import numpy as np
    ext_suffix = sysconfig.get_config_var('EXT_SUFFIX')
    inc_py = sysconfig.get_paths()['include']
    for mod, src in SRCS.items():
        so = os.path.join(OUT, mod + ext_suffix)
        src_path = os.path.join(REF, src)
        if (not force) and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src_path):
            continue
        with open(src_path) as f:
            text = _patch(f.read())
        pyx = os.path.join(OUT, mod + '.pyx')
        with open(pyx, 'w') as f:
            f.write(text)
        c_file = os.path.join(OUT, mod + '.c')
        subprocess.check_call([sys.executable, '-m', 'cython', '-3', '--directive', 'language_level=3',
                               pyx, '-o', c_file])
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-Wno-cpp',
                               '-Wno-unused-function', '-I', inc_py, '-I', np.get_include(),
                               c_file, '-o', so])
    build_affine(force)
    return True


AFFINE_CU = '/root/reference/lib/ops/affine_channel_nd_op.cu'
AFFINE_SO = os.path.join(OUT, 'libref_affine.so')


def build_affine(force=False):
    """hipcc (cross-compiles gfx950 without a GPU) on the reference .cu where it lies; False when the reference is absent."""
    if not os.path.isfile(AFFINE_CU):
        return False
    os.makedirs(OUT, exist_ok=True)
    drv = os.path.join(HERE, 'ref_affine', 'ref_affine_driver.hip')
    shim = os.path.join(HERE, 'ref_affine', 'shim')
    deps = [AFFINE_CU, drv, os.path.join(shim, 'caffe2', 'core', 'context.h')]
    if (not force) and os.path.exists(AFFINE_SO) and all(os.path.getmtime(AFFINE_SO) >= os.path.getmtime(d) for d in deps):
        return True
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip', '-I', shim,
                           '-DREF_AFFINE_CU="%s"' % AFFINE_CU, drv, '-o', AFFINE_SO])
    return True


def load_affine():
    """ctypes handle of oracle/_ref/libref_affine.so with argtypes set, or None when it was never built."""
    import ctypes as C
    if not os.path.exists(AFFINE_SO):
        return None
    lib = C.CDLL(AFFINE_SO)
    lib.ref_affine_channel_nd_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_char_p, C.c_int]
    lib.ref_affine_channel_nd_bwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_char_p, C.c_int]
    lib.ref_affine_channel_nd_fwd.restype = lib.ref_affine_channel_nd_bwd.restype = C.c_int
    return lib


def load():
    """Import the built reference modules (or return None if absent)."""
    import importlib
    if OUT not in sys.path:
        sys.path.insert(0, OUT)
    try:
        return (importlib.import_module('ref_cython_nms'), importlib.import_module('ref_cython_bbox'))
    except ImportError:
        return None


if __name__ == '__main__':
    print('built' if build(force='--force' in sys.argv) else 'reference not present; skipped')
    print(load())
    print(load_affine())
