#!/usr/bin/env python3
"""What makes contexts created later in a process decode at half their rate?  Decodes the headline
batch (64 images at 368 x 368) with four streams, two contexts in turn on each, and prints images/s,
after one of several things has been done in the process first:

  --touch pin | d2h | event | wait2 | record   pinned memory, an async copy to it, Event.synchronize, a torch stream
                                               waiting for another torch stream, an event recorded on a context's stream
  --touch wait        a torch stream waits for an event recorded on a context's stream; the context is then destroyed
  --touch wait_keep   the same, the context stays alive
  --threads-first     three contexts decode through their launcher threads (pd_submit) and are destroyed
  --gather-first [--variant skeleton|object|full]   ... with skeleton buffers / a SkeletonGather made / used
  --torch-pool-first  torch streams created and used before the contexts are made

Usage: GPU_MAX_HW_QUEUES=8 python tools/stream_wait_probe.py [options] [--steps N]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--torch-pool-first', action='store_true')
    ap.add_argument('--steps', type=int, default=600)
    ap.add_argument('--touch', default='', help='comma list of things done once before the contexts are made: d2h (async copy to pinned memory), event (Event.synchronize), wait (a torch stream waits for an external stream), pin (pin_memory only)')
    ap.add_argument('--threads-first', action='store_true', help='three contexts decode through their launcher threads (pd_submit) and are closed, before the measurement')
    ap.add_argument('--variant', default='full', help='of --gather-first: skeleton (buffers only, no gather object), object (gather object made, never used), full')
    ap.add_argument('--gather-first', action='store_true', help='a SkeletonGather (torch copy stream, pinned buffers) is used and dropped before the measurement')
    args = ap.parse_args()
    import torch
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    B, H, W = 64, 368, 368
    heat, paf = synth.batch(B, H, W, (1, 8), seed0=0, paf_order='xy')
    d_heat = [torch.from_numpy(np.roll(heat, -k, axis=0)).cuda() for k in range(8)]
    d_paf = [torch.from_numpy(np.roll(paf, -k, axis=0)).cuda() for k in range(8)]
    keep = None
    if args.torch_pool_first:
        keep = [torch.cuda.Stream() for _ in range(2)]
        x = torch.zeros(1024, device='cuda')
        for st in keep:      # (a stream takes its hardware queue when it is first used)
            with torch.cuda.stream(st):
                x.add_(1)
        torch.cuda.synchronize()
    g = None
    for what in [w for w in args.touch.split(',') if w]:
        x = torch.zeros(1 << 20, dtype=torch.uint8, device='cuda')
        if what == 'pin':
            hp = torch.zeros(1 << 20, dtype=torch.uint8).pin_memory()
        elif what == 'd2h':
            hp = torch.zeros(1 << 20, dtype=torch.uint8).pin_memory()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                hp.copy_(x, non_blocking=True)
            torch.cuda.synchronize()
        elif what == 'event':
            ev = torch.cuda.Event()
            x.add_(1)
            ev.record()
            ev.synchronize()
        elif what == 'wait2':       # two torch streams, no stream of the library's
            s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
            with torch.cuda.stream(s1):
                x.add_(1)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                x.add_(1)
            torch.cuda.synchronize()
        elif what == 'record':      # an event recorded on a stream of the library's, nobody waits for it
            tmp = DecodeEngine(max_batch=1, max_h=64, max_w=64, mode='D')
            ext = torch.cuda.ExternalStream(int(tmp.ctx.stream))
            ev = torch.cuda.Event()
            ev.record(ext)
            torch.cuda.synchronize()
            tmp.close()
        elif what == 'wait_keep':   # as wait, but the waited-for context stays alive
            keep_ctx = DecodeEngine(max_batch=1, max_h=64, max_w=64, mode='D')
            ext = torch.cuda.ExternalStream(int(keep_ctx.ctx.stream))
            st = torch.cuda.Stream()
            st.wait_stream(ext)
            torch.cuda.synchronize()
        elif what == 'wait_rev':    # a stream of the library's waits for a torch stream: not possible through torch alone
            pass
        elif what == 'wait':
            import ctypes
            tmp = DecodeEngine(max_batch=1, max_h=64, max_w=64, mode='D')
            ext = torch.cuda.ExternalStream(int(tmp.ctx.stream))
            st = torch.cuda.Stream()
            st.wait_stream(ext)
            torch.cuda.synchronize()
            tmp.close()
    kw = dict(max_batch=B, max_h=H, max_w=W, mode='D', max_peaks_per_part=128, max_persons=128, max_pairs_per_image=40000)
    if args.threads_first or args.gather_first:
        from poseestimation_amd import sharding
        pre = [DecodeEngine(**kw) for _ in range(3)]
        if args.gather_first:
            if args.variant == 'skeleton':
                class G:
                    capacity = 1 << 20
                    buf = torch.zeros(2 * 6 * (1 << 20), dtype=torch.uint8, device='cuda')
                    def slice_ptr(self, s, j):
                        return self.buf.data_ptr() + (s * 6 + j) * self.capacity
                    def start(self, s):
                        pass
                    def finish(self, s):
                        torch.cuda.synchronize()
                g = G()
            else:
                g = sharding.SkeletonGather([e.ctx for e in pre], 1 << 20, depth=2, local=True)
                if args.variant == 'object':
                    g.start = lambda s: None
                    g.finish = lambda s: torch.cuda.synchronize()
            for e in pre:
                e.ctx.set_host_results(False)
        for it in range(20):
            for j, e in enumerate(pre):
                if args.gather_first:
                    if args.threads_first:
                        e.ctx.submit(d_heat[j].data_ptr(), d_paf[j].data_ptr(), B, H // 8, W // 8, H, W, 0.1, 0.05, g.slice_ptr(it % 2, j), g.capacity)
                    else:
                        e.ctx.set_skeleton_buffer(g.slice_ptr(it % 2, j), g.capacity)
                        e.enqueue_device(d_heat[j].data_ptr(), d_paf[j].data_ptr(), B, H // 8, W // 8)
                elif args.threads_first:
                    e.ctx.submit(d_heat[j].data_ptr(), d_paf[j].data_ptr(), B, H // 8, W // 8, H, W, 0.1, 0.05, 0, 0)
            for e in pre:
                e.ctx.flush()
            if args.gather_first:
                g.start(it % 2)
                g.finish(it % 2)
            else:
                for e in pre:
                    e.ctx.fetch()
        torch.cuda.synchronize()
        for e in pre:
            e.close()
        g = None
    engs = [DecodeEngine(**kw) for _ in range(4)]
    engs += [DecodeEngine(share_stream_with=engs[i], **kw) for i in range(4)]
    import gc
    gc.collect()
    gc.freeze()

    def run(n):
        pending = []
        for i in range(n):
            e = engs[i % 8]
            if len(pending) == 8:
                pending.pop(0).ctx.fetch()
            e.enqueue_device(d_heat[i % 8].data_ptr(), d_paf[i % 8].data_ptr(), B, H // 8, W // 8)
            pending.append(e)
        for e in pending:
            e.ctx.fetch()

    run(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('torch pool first: %s, threads first: %s, gather first: %s, PD_STREAM_PRIORITY=%s: %.0f images/s (%.1f us per decode)'
          % (args.torch_pool_first, args.threads_first, (args.gather_first, args.variant, args.touch), os.environ.get('PD_STREAM_PRIORITY', '0'), B * args.steps / dt, dt / args.steps * 1e6))
    del keep


if __name__ == '__main__':
    main()
