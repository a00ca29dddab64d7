#!/usr/bin/env python3
"""Decode throughput benchmark: network outputs (19 heatmaps + 38 PAFs at stride 8) in
HBM -> per-person skeletons on the host, on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" decodes one batch of synthetic network outputs per GPU (weak scaling: the
per-GPU batch is fixed).  For N > 1 launch under torch.distributed.run; each rank owns
one GPU and one HIP stream, images are sharded with no data-path collective and the
skeleton lists are gathered to rank 0 (one RCCL gather per step).  Rank 0 prints one
JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (batch per GPU, H, W, people per image, description)
    'b64_368': (64, 368, 368, (1, 8), 'synthetic batch 64 at 368x368 (46x46x57 network output), single scale'),
    'b256_656': (256, 656, 656, 30, 'synthetic batch 256 at 656x656, ~30 people per image'),
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='b64_368', choices=sorted(WORKLOADS))
    ap.add_argument('--mode', default='D', choices=['P', 'D'])
    ap.add_argument('--cpu-images', type=int, default=12, help='images timed on the CPU oracle (rank 0, N=1)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--people', type=int, default=0, help='experiments: fixed people per image')
    ap.add_argument('--contexts', type=int, default=0,
                    help='decode contexts (HIP streams) per GPU, used round-robin (0: 4 on one GPU, 3 per rank under torch.distributed)')
    ap.add_argument('--pipeline', default='fetch', choices=['fetch', 'rounds'],
                    help='N = 1 only.  fetch: every decode replays its launch graph and copies its full packed '
                         'results (with the connection lists) to the host; rounds: the N > 1 pipeline with a '
                         'plain device-to-host copy in place of the RCCL gather (peaks + person rows per round '
                         'of decodes, eager launches from the contexts\' launcher threads)')
    ap.add_argument('--hw-queues', type=int, default=-1,
                    help='GPU_MAX_HW_QUEUES for this process when the environment asks for fewer '
                         '(0: leave it; -1: 8 on one GPU, left alone under torch.distributed)')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # PD_BENCH_FORCE_DIST=1 runs the N > 1 code path (RCCL gather of the skeleton buffers)
    # with a single rank, which is how that path is rehearsed on a one-GPU box
    use_dist = world > 1 or os.environ.get('PD_BENCH_FORCE_DIST') == '1'
    rounds = use_dist or args.pipeline == 'rounds'
    if args.contexts <= 0:
        args.contexts = 4
    if args.hw_queues < 0:
        # measured: replayed graphs want a queue per context (4 contexts: 8 queues), the eager
        # launches of the round pipeline lose throughput with more than the default 4
        args.hw_queues = 0 if rounds else 8

    # HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default);
    # streams that share a queue run one behind the other.  Every decode context needs a
    # queue of its own next to the one torch uses, so raise the count before HIP starts
    # (replayed launch graphs: four contexts on eight queues decode 12 % more than three on four).
    try:
        have_q = int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        have_q = 4
    if args.hw_queues > have_q:
        os.environ['GPU_MAX_HW_QUEUES'] = str(args.hw_queues)
    import torch
    import torch.distributed as dist
    from poseestimation_amd import _ffi, sharding, synth
    from poseestimation_amd.predict_decode import DecodeEngine, Results

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    if use_dist:
        # RCCL writes a version banner to stdout when its first communicator comes up; stdout
        # carries the JSON line only, so the banner is sent to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if world > 1:
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            else:
                dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', rank=0, world_size=1,
                                        device_id=torch.device('cuda', local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    B, H, W, people, desc = WORKLOADS[args.workload]
    if args.people:      # experiments only: a fixed number of people per image
        people = args.people
        desc += ' (people per image overridden: %d)' % people
    h, w = H // 8, W // 8
    heat, paf = synth.batch(B, H, W, people, seed0=rank * B, paf_order='xy' if args.mode == 'D' else 'yx')
    d_heat = torch.from_numpy(heat).cuda()
    d_paf = torch.from_numpy(paf).cuda()
    torch.cuda.synchronize()

    # several contexts (one HIP stream each) used round-robin: while the skeletons of
    # batch i are copied out and gathered, the kernels of the next batches already run,
    # and the short kernels of different batches overlap on the chip
    engs = [DecodeEngine(max_batch=B, max_h=H, max_w=W, mode=args.mode, device_id=local_rank,
                         max_peaks_per_part=128, max_persons=128, max_pairs_per_image=40000)
            for _ in range(args.contexts)]
    eng = engs[0]
    gather_cap = 1 << 18      # per rank and step: header, count tables, peaks, person rows
    while gather_cap < B * 4096:
        gather_cap <<= 1

    # (contexts that are given a skeleton buffer launch eagerly: graph replay next to a live
    # process group faulted on this stack, see pd_set_skeleton_buffer)
    GATHER_DEPTH = 2   # decodes per context between two gathers
    gather = gather1 = None
    if rounds:
        gather = sharding.SkeletonGather([e.ctx for e in engs], gather_cap, depth=GATHER_DEPTH, local=not use_dist)
        gather1 = sharding.SkeletonGather([engs[0].ctx], gather_cap, depth=GATHER_DEPTH, local=True)
        for e in engs:      # everything a rank produces leaves through the gather
            e.ctx.set_host_results(False)

    host_t = [0.0, 0.0, 0.0, 0.0, 0]   # finish, submit, flush, start, rounds

    def run(nsteps, pool=None):
        """nsteps steps = nsteps batches decoded and their skeletons brought to the host
        (N = 1: the packed results of the batch; N > 1: every rank's peaks and person
        rows gathered to rank 0 over RCCL)."""
        n, last = len(engs), None
        pool_arg = pool
        if gather is None:
            pool = pool or engs
            n = len(pool)
            pending = []
            for i in range(nsteps):
                e = pool[i % n]
                if len(pending) == n:
                    last = pending.pop(0).ctx.fetch()   # waits for that stream; results now on the host
                e.enqueue_device(d_heat.data_ptr(), d_paf.data_ptr(), B, h, w, (H, W))
                pending.append(e)
            while pending:
                last = pending.pop(0).ctx.fetch()
            return last
        # rounds of GATHER_DEPTH decodes per context; the gather of round r (set r % 2 of the
        # send buffers) is in flight while round r + 1 decodes; the host never waits for a decode
        g, pool = gather, engs
        if pool_arg is not None:
            g, pool = gather1, pool_arg
        n = len(pool)
        last, r, done = None, 0, 0
        pending = [False, False]
        while done < nsteps:
            s = r % 2
            ta = time.perf_counter()
            if pending[s]:
                rows = g.finish(s)
                last = rows if rows is not None else last
            tb = time.perf_counter()
            k = min(n * GATHER_DEPTH, nsteps - done)
            # the contexts' launcher threads issue the (eager) kernel launches in parallel
            for j in range(k):
                pool[j % n].ctx.submit(d_heat.data_ptr(), d_paf.data_ptr(), B, h, w, H, W, 0.1, 0.05,
                                       g.slice_ptr(s, j), g.capacity)
            tc = time.perf_counter()
            for e in pool:
                e.ctx.flush()
            td = time.perf_counter()
            g.start(s)
            te = time.perf_counter()
            host_t[0] += tb - ta
            host_t[1] += tc - tb
            host_t[2] += td - tc
            host_t[3] += te - td
            host_t[4] += 1
            pending[s] = True
            done += k
            r += 1
        for s in (r % 2, (r + 1) % 2):
            if pending[s]:
                rows = g.finish(s)
                last = rows if rows is not None else last
        return last

    def barrier():
        torch.cuda.synchronize()         # device-wide: covers the contexts' own streams
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: only the roofline kernel carries an event pair (one pair per step);
    # timing every kernel would put two dozen event records between the kernels of
    # every step and slow the job down by a third.  The full per-stage breakdown comes
    # from a separate, untimed pass afterwards.
    ROOF_STAGE = 'filter'
    # Python's cyclic collector walks every object the imports left behind (torch, numpy) at its
    # first full pass: a 35 ms stop some 650 steps into the run.  Those objects are parked in
    # the permanent generation, as a long-running server would do after start-up.
    import gc
    gc.collect()
    gc.freeze()
    for e in engs:
        e.ctx.profile(stages=[ROOF_STAGE])
    # every context reaches its captured launch graph (third identical call); with a process
    # group, the first gathers also set up RCCL's channels and the launcher threads
    run(3 * len(engs) if gather is None else 8 * len(engs) * GATHER_DEPTH)
    last = run(args.warmup)
    for e in engs:
        e.ctx.profile_read()
    barrier()
    host_t[:] = [0.0, 0.0, 0.0, 0.0, 0]
    t0 = time.perf_counter()
    last = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get('PD_BENCH_TIMERS') == '1' and host_t[4]:
        sys.stderr.write('host us per round: finish %.1f submit %.1f flush %.1f start %.1f (rounds %d)\n'
                         % tuple([x / host_t[4] * 1e6 for x in host_t[:4]] + [host_t[4]]))
    roof_ms, roof_n = 0.0, 0
    for e in engs:
        v = e.ctx.profile_read()[ROOF_STAGE]
        roof_ms += v[0]
        roof_n += v[1]
    # the same event pair with one context decoding alone: the kernel's own duration, which is
    # what a kernel trace (rocprofv3 serialises the kernels of all streams) reports; in the
    # timed region the kernel shares the chip with the kernels of the other contexts' decodes
    alone_ms, alone_n = 0.0, 0
    if not use_dist:
        run(min(100, args.steps), pool=engs[:1])
        alone_ms, alone_n = engs[0].ctx.profile_read()[ROOF_STAGE]
    prof = {}
    stage_steps = max(10, min(50, args.steps))
    for e in engs:
        e.ctx.profile(True)
    run(3 * len(engs))
    for e in engs:
        e.ctx.profile_read()
    run(stage_steps)
    for e in engs:
        for k, v in e.ctx.profile_read().items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + v[0], a[1] + v[1])
        e.ctx.profile(False)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    parity_failed = False
    if rank == 0:
        images = world * B * args.steps
        n, N = h * w, H * W
        # Roofline kernel: k_cell_filter, the float32 evaluation of the smoothed x8 map
        # and its non-maximum suppression (with k_cells, which picks the 8x8 cells it has
        # to visit, it stands for the reference's upsample + Gaussian + NMS pass over every
        # pixel of the 18 part channels).  Algorithmic bytes per image of that pass:
        # upsample read 18*n*4 + write 18*N*4, Gaussian read + write 2*18*N*4, NMS read
        # 18*N*4  =  72*n + 288*N.  The kernel itself touches a few percent of that, so the
        # fraction says how much faster than a materialising implementation at HBM peak
        # the stage runs, not how busy HBM is.
        kname = 'k_cell_filter'
        sm_ms, sm_n = roof_ms, roof_n
        if sm_n == 0:       # the exact streaming kernel ran alone (no filter tables for this geometry)
            kname = 'k_smooth_nms_stream'
            sm_ms, sm_n = prof['smooth_nms']
        alg_bytes = B * (72.0 * n + 288.0 * N)
        avg_s = (sm_ms / max(sm_n, 1)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_%s.json' % args.workload)
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                traffic = json.load(f).get('roofline_kernel_hbm_bytes_per_launch')
        # the whole decode against the whole-path byte count per image:
        # 228*n + 228*N + 216*N (+ 80 bytes per candidate pair, not counted here)
        job_bytes = 228.0 * n + 444.0 * N
        value = images / dt
        out = {
            'metric': 'decoded images/sec (heatmap+PAF->skeletons) at %dx%d' % (H, W),
            'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': desc, 'batch_per_gpu': B, 'mode': args.mode,
                       'contexts_per_gpu': len(engs), 'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'default'),
                       'people_per_image': list(people) if isinstance(people, tuple) else people,
                       'outputs': 'full packed results (peaks, connection lists, person rows) copied to the host every step'
                                  if gather is None else
                                  'peaks and person rows copied to the host once per round of %d decodes' % gather.n
                                  if not use_dist else
                                  'peaks and person rows of every rank gathered to rank 0 (RCCL) and copied to its host, '
                                  'once per round of %d decodes per rank' % gather.n},
            'roofline': {'bound': 'hbm', 'kernel': kname, 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'avg_launch_ms': avg_s * 1e3, 'launches': sm_n,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'measured': 'HIP event pair around the kernel in every step of the timed region, '
                                     '%d decodes in flight on the chip' % len(engs),
                         'note': 'algorithmic bytes are those of the reference\'s materialising upsample + Gaussian + NMS '
                                 'pass, which this kernel replaces on the few percent of the map that can hold a peak: '
                                 'frac > 1 means faster than that pass could run at HBM peak; the kernel itself is '
                                 'bound by vector-ALU issue (profiles/*sq_counters*), its HBM traffic is `traffic`'},
            'job_hbm_frac_per_gpu': (value / world) * job_bytes / 1e9 / HBM_PEAK_GBS,
            'stage_ms_per_launch_untimed_pass': {k: v[0] / v[1] for k, v in prof.items() if v[1]},
        }
        if alone_n and kname == 'k_cell_filter':
            a_s = alone_ms / alone_n * 1e-3
            out['roofline']['one_context'] = {
                'avg_launch_ms': a_s * 1e3, 'launches': alone_n, 'achieved': alg_bytes / a_s / 1e9,
                'frac': alg_bytes / a_s / 1e9 / HBM_PEAK_GBS,
                'measured': 'same event pair, one context decoding alone after the timed region '
                            '(comparable with a kernel trace, which serialises kernels)'}
        res0 = last if gather is None else sharding.SkeletonGather.results(last)[0]
        out['persons_last_step'] = int(res0['person_count'].sum())
        out['filter_stats_last_step'] = eng.ctx.filter_stats()

        if not use_dist and not args.no_cpu:
            from oracle import decode_ref as ref
            ncpu = min(args.cpu_images, B)
            r = Results(res0, eng.mode)
            tc = time.perf_counter()
            ok = True
            for b in range(ncpu):
                hf = ref.resize_cubic(heat[b], fx=8, fy=8).astype(np.float64)
                pf = ref.resize_cubic(paf[b], fx=8, fy=8).astype(np.float64)
                peaks = ref.get_peaks(hf, 0.1, args.mode, use_scipy=True)
                conns, special = ref.get_connections(peaks, pf, 0.05, args.mode)
                subset = ref.multi_person_parse(peaks, conns, special, args.mode)
                g = r.subset(b)
                ok = ok and g.shape == subset.shape and np.array_equal(g[:, :18], subset[:, :18]) \
                    and np.allclose(g[:, 18:], subset[:, 18:], rtol=1e-5, atol=0)
            tcpu = time.perf_counter() - tc
            out['cpu_baseline'] = {'value': ncpu / tcpu, 'unit': 'images/s', 'cores': 1, 'kind': 'port',
                                   'sample': 'first %d images of the same batch through oracle/decode_ref.py '
                                             '(numpy bicubic + scipy gaussian_filter + numpy decode), one thread; '
                                             'host has %d cpus' % (ncpu, os.cpu_count())}
            out['parity_checked_images'] = ncpu
            out['parity_ok'] = bool(ok)
        print(json.dumps(out))
        sys.stdout.flush()
        parity_failed = out.get('parity_ok') is False
    for e in engs:
        e.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and parity_failed:
        raise SystemExit('bench: the decode differs from the CPU oracle (parity_ok false)')


if __name__ == '__main__':
    main()
