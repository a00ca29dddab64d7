#!/usr/bin/env python3
"""Where k_parse_pack's time goes (GPU).  Loads the stamped measurement build
(make -C poseestimation_amd/csrc stamps; PD_LIBRARY points at it), decodes the headline batch
a few times on one context and prints, per phase, the median over the workgroups of the last
launch in microseconds from the first workgroup's start: drawing the image number, the assembly
wavefront, the three copy wavefronts, the look-back over the row counts of the images before,
and the writing of the person rows.

Usage: PD_LIBRARY=poseestimation_amd/libposedecode_stamps.so python tools/parse_pack_split.py [B H W]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    import torch
    from poseestimation_amd import _ffi, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 368, 368)
    L = _ffi.lib()
    L.pd_debug_pp_stamps.argtypes = [C.c_void_p, C.c_int]
    outs = [synth.scene(H, W, b % 7, 1000 + b) for b in range(B)]
    heat = torch.from_numpy(np.stack([o[0] for o in outs])).cuda()
    paf = torch.from_numpy(np.stack([o[1] for o in outs])).cuda()
    eng = DecodeEngine(B, H, W)
    for it in range(5):
        eng.enqueue_device(heat.data_ptr(), paf.data_ptr(), B, H // 8, W // 8)
        res = eng.fetch()
    st = np.zeros((B, 16), np.int64)
    assert L.pd_debug_pp_stamps(st.ctypes.data, B) == 0
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    names = ['start', 'image number drawn', 'row count known and published (assembly done)', 'copy wavefront 1 done',
             'copy wavefront 2 done', 'copy wavefront 3 done', 'look-back done', 'rows (last image: header) written']
    print('k_parse_pack, batch %d at %dx%d, %d persons: microseconds after the first workgroup starts'
          % (B, H, W, int(res.packed['person_count'].sum())))
    names += ['connections staged', 'limbs applied']
    for q, n in enumerate(names):
        if not n:
            continue
        col = us[:, q]
        print('  %-48s min %6.2f  median %6.2f  max %6.2f' % (n, col.min(), np.median(col), col.max()))
    lim = (st[:, 9] - st[:, 8]) / 100.0
    print('  applying the limbs                               min %6.2f  median %6.2f  max %6.2f' % (lim.min(), np.median(lim), lim.max()))
    own = (st[:, 2] - st[:, 1]) / 100.0
    print('  assembly wavefront, own duration                 min %6.2f  median %6.2f  max %6.2f' % (own.min(), np.median(own), own.max()))
    cp = (st[:, 3:6].max(axis=1) - st[:, 1]) / 100.0
    print('  copy wavefronts, own duration (slowest of 3)     min %6.2f  median %6.2f  max %6.2f' % (cp.min(), np.median(cp), cp.max()))
    lb = (st[:, 6] - st[:, 2]) / 100.0
    print('  look-back, wait after publishing                 min %6.2f  median %6.2f  max %6.2f' % (lb.min(), np.median(lb), lb.max()))
    wr = (st[:, 7] - st[:, 6]) / 100.0
    print('  barrier + writing the rows                       min %6.2f  median %6.2f  max %6.2f' % (wr.min(), np.median(wr), wr.max()))


if __name__ == '__main__':
    main()
