#!/usr/bin/env python3
"""Decode throughput benchmark: network outputs (19 heatmaps + 38 PAFs at stride 8) in
HBM -> per-person skeletons on the host, on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

Workloads (the configurations of BASELINE.json):
  b64_368           batch 64 at 368x368, single scale -- the configuration the metric is quoted on
  b256_656          batch 256 at 656x656, ~30 people per image
  ms4_b128_368      batch 128, four test scales (0.5, 1, 1.5, 2) averaged before the decode
  stream_b1024_368  1024 images of 46x46x57 per step, streamed in chunks of 256 through the
                    round pipeline (skeleton buffers gathered once per round)

One "step" decodes one batch (ms4, b64, b256) or one pass of 1024 images (stream) per GPU:
weak scaling, the per-GPU work is fixed.  For N > 1 launch under torch.distributed.run; each
rank owns one GPU, images are sharded with no data-path collective and the skeleton lists
are gathered to rank 0 (one RCCL gather per round of decodes).  Rank 0 prints one JSON line.
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
    # name: batch per decode, H, W, people per image, decodes per step, default steps, description
    'b64_368': dict(B=64, H=368, W=368, people=(1, 8), per_step=1, steps=2000, kind='single',
                    desc='synthetic batch 64 at 368x368 (46x46x57 network output), single scale'),
    'b256_656': dict(B=256, H=656, W=656, people=30, per_step=1, steps=300, kind='single',
                     desc='synthetic batch 256 at 656x656, ~30 people per image'),
    'ms4_b128_368': dict(B=128, H=368, W=368, people=(1, 8), per_step=1, steps=30, kind='multiscale',
                         desc='synthetic batch 128 at 368x368, four test scales (0.5, 1, 1.5, 2: network outputs of '
                              '23^2, 46^2, 69^2, 92^2 x 57) resized, averaged and decoded'),
    # (chunks of 256: one rank's share of 1024 images is decoded in as few launches as keep three contexts busy;
    # measured in the round pipeline, images/s by chunk size: 64 1.17 M, 128 1.47 M, 256 1.57 M, 512 1.59 M;
    # PD_BENCH_CHUNK=n overrides)
    'stream_b1024_368': dict(B=256, H=368, W=368, people=(1, 8), per_step=4, steps=125, kind='single',
                             desc='1024 synthetic images of 46x46x57 per step, streamed in 4 chunks of 256 through the '
                                  'round pipeline'),
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
SCALES = (0.5, 1.0, 1.5, 2.0)


def graph_friendly_buffers(nbuf, n_ctx, slots=6):
    """Largest k <= nbuf such that a context used round-robin sees at most `slots` distinct
    input buffers (a context keeps eight recorded launch sequences)."""
    import math
    for k in range(nbuf, 0, -1):
        if k // math.gcd(k, n_ctx) <= slots:
            return k
    return 1


def _cpu_image(job):
    """Worker of cpu_all_cores: one image through the CPU restatement; returns its person count."""
    heat, paf, mode = job
    from oracle import decode_ref as ref
    hf = ref.resize_cubic(heat, fx=8, fy=8).astype(np.float64)
    pf = ref.resize_cubic(paf, fx=8, fy=8).astype(np.float64)
    peaks = ref.get_peaks(hf, 0.1, mode, use_scipy=True)
    conns, special = ref.get_connections(peaks, pf, 0.05, mode)
    return int(ref.multi_person_parse(peaks, conns, special, mode).shape[0])


def cpu_all_cores(args, wl):
    """The CPU restatement on all the cores this job has (images are independent: one process
    per core, one image at a time each).  Runs before the process has touched the GPU, because
    its workers are forked.  A reported figure beside `cpu_baseline`; None if anything goes wrong."""
    import multiprocessing as mp
    pool = None
    try:
        from oracle import decode_ref   # noqa: F401  (loaded before the fork)
        from poseestimation_amd import synth
        try:
            share = len(os.sched_getaffinity(0))
        except AttributeError:
            share = os.cpu_count() or 1
        # a GPU box grants 16 cores of a much larger machine and says so through OMP_NUM_THREADS
        workers = max(1, min(share, int(os.environ.get('OMP_NUM_THREADS', '16') or 16), 16))
        per = 2
        n = workers * per
        heat, paf = synth.batch(min(n, wl['B']), wl['H'], wl['W'], wl['people'], seed0=0,
                                paf_order='xy' if args.mode == 'D' else 'yx')
        jobs = [(heat[i % len(heat)], paf[i % len(heat)], args.mode) for i in range(n)]
        pool = mp.get_context('fork').Pool(workers)
        pool.map_async(_cpu_image, jobs[:workers], chunksize=1).get(timeout=120)   # every worker warm
        t0 = time.perf_counter()
        pool.map_async(_cpu_image, jobs, chunksize=1).get(timeout=240)
        dt = time.perf_counter() - t0
        return {'value': n / dt, 'unit': 'images/s', 'cores': workers, 'kind': 'port',
                'sample': '%d images of the workload through oracle/decode_ref.py, %d single-threaded worker '
                          'processes at once' % (n, workers)}
    except Exception as exc:      # the figure is optional: never let it stop the bench
        sys.stderr.write('cpu_all_cores skipped: %r\n' % (exc,))
        return None
    finally:
        if pool is not None:
            pool.terminate()
            pool.join()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=0, help='timed steps (0: the workload\'s default)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='b64_368', choices=sorted(WORKLOADS))
    ap.add_argument('--mode', default='D', choices=['P', 'D'])
    ap.add_argument('--cpu-images', type=int, default=12, help='images timed on the CPU oracle (rank 0, N=1)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the untimed extra passes (per-stage times, dense kernel, host-resident inputs)')
    ap.add_argument('--people', type=int, default=0, help='experiments: fixed people per image')
    ap.add_argument('--no-stage-timing', action='store_true',
                    help='experiments: no event pair around the roofline kernel in the timed region')
    ap.add_argument('--contexts', type=int, default=0,
                    help='decode contexts (HIP streams) per GPU, used round-robin (0: 4; 3 in the round pipeline and for the multi-scale workload)')
    ap.add_argument('--depth', type=int, default=0,
                    help='fetch pipeline: contexts per stream, used in turn, so that a stream holds that many '
                         'decodes (0: 2 for maps of at most 4096 pixels, else 1; the round pipeline keeps its own depth)')
    ap.add_argument('--pipeline', default='', choices=['', 'fetch', 'rounds'],
                    help='N = 1 only.  fetch: every decode copies its full packed results (with the connection '
                         'lists) to the host; rounds: the N > 1 pipeline with a plain device-to-host copy in place of '
                         'the RCCL gather (peaks + person rows per round of decodes, issued by the contexts\' '
                         'launcher threads).  Default: fetch, rounds for the stream workload')
    ap.add_argument('--hw-queues', type=int, default=-1,
                    help='GPU_MAX_HW_QUEUES for this process when the environment asks for fewer '
                         '(0: leave it; -1: 8)')
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if os.environ.get('PD_BENCH_CHUNK'):      # comparison runs: the same images per step in chunks of another size
        c = int(os.environ['PD_BENCH_CHUNK'])
        wl['desc'] = wl['desc'].replace('%d chunks of %d' % (wl['per_step'], wl['B']), '%d chunks of %d' % (wl['B'] * wl['per_step'] // c, c))
        wl['per_step'] = wl['B'] * wl['per_step'] // c
        wl['B'] = c
    if args.steps <= 0:
        args.steps = wl['steps']
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # PD_BENCH_FORCE_DIST=1 runs the N > 1 code path (RCCL gather of the skeleton buffers)
    # with a single rank, which is how that path is rehearsed on a one-GPU box
    use_dist = world > 1 or os.environ.get('PD_BENCH_FORCE_DIST') == '1'
    multiscale = wl['kind'] == 'multiscale'
    if not args.pipeline:
        args.pipeline = 'rounds' if wl['per_step'] > 1 else 'fetch'
    rounds = (use_dist or args.pipeline == 'rounds') and not multiscale
    # The chip serves four hardware queues at a time: the fetch pipeline runs four contexts, the
    # round pipeline three next to the stream of its gather / copy (measured: rounds with four
    # contexts 0.77 M images/s on four queues and 0.65 M on eight, with three on eight 1.0 M).
    if args.contexts <= 0:
        args.contexts = 3 if (multiscale or rounds) else 4
    if args.hw_queues < 0:
        args.hw_queues = 8

    # HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default);
    # streams that share a queue run one behind the other.  Every decode context needs a
    # queue of its own next to the one torch uses, so raise the count before HIP starts.
    try:
        have_q = int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        have_q = 4
    if args.hw_queues > have_q:
        os.environ['GPU_MAX_HW_QUEUES'] = str(args.hw_queues)
    # (sharing device memory between the ranks' processes needs dmabuf handles on this stack; the launcher's environment
    # normally carries this already)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    all_cores = None
    # (not under a profiler: its preloaded library may have started the GPU runtime already,
    # and a process in that state is not forked)
    profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    if world == 1 and not use_dist and not multiscale and not args.no_cpu and not args.no_extras and not profiled:
        all_cores = cpu_all_cores(args, wl)     # before anything touches the GPU
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    # PD_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path where the ranks cannot have a GPU each (RCCL refuses two
    # ranks on one GPU): the ranks share the GPUs there are, the gathers go through host memory.  Its rates mean nothing.
    backend = os.environ.get('PD_BENCH_BACKEND', 'nccl')
    if backend == 'gloo':
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if use_dist and backend == 'gloo':
        dist.init_process_group('gloo')
        dist.barrier()
    elif use_dist:
        # RCCL writes a version banner to stdout when its first communicator comes up; stdout
        # carries the JSON line only, so the banner is sent to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ:
                # (a one-rank rehearsal under torch.distributed.run must use the launcher's store
                # too: its workers connect to that store whatever address they are given)
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

    env = dict(args=args, wl=wl, world=world, rank=rank, local_rank=local_rank, use_dist=use_dist, rounds=rounds,
               torch=torch, dist=dist, all_cores=all_cores, comm_device='cpu' if backend == 'gloo' else 'cuda',
               token=torch.zeros(1, device='cpu' if backend == 'gloo' else 'cuda') if use_dist else None)
    out = run_multiscale(env) if multiscale else run_single(env)
    parity_failed = False
    if rank == 0 and world == 1 and not use_dist and not args.no_extras and not profiled and args.workload == 'b64_368' \
            and not args.people and os.environ.get('PD_BENCH_NO_OTHERS') != '1':
        # the default run also times short regions of the other configurations, the headline batch
        # in the other decode variant and one image through detect()
        others_failed = other_workloads(env, out)
        parity_failed = parity_failed or others_failed
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
        parity_failed = parity_failed or out.get('parity_ok') is False
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit('bench: the decode differs from the CPU oracle (parity_ok false)')


def other_workloads(env, out):
    """Short timed regions of the configurations the headline line does not carry (BASELINE
    configs 3, 4, 5), the headline batch under utils/postprocess.py semantics (mode P: the half of
    the oracle that is pinned by the reference's own outputs) and a single image through
    detect() (config 1's shape).  Fills out['other_workloads'], out['mode_p'] and
    out['detect_single_image']; returns True if any parity check failed."""
    import copy
    args = env['args']
    failed = False
    legs = {}
    # the headline batch in mode P.  (First: the round pipeline of the stream leg orders its copies behind the decodes
    # with stream waits, and once contexts whose streams were waited for are destroyed, contexts made later in the
    # process were seen to decode at half the rate -- the same batch at 0.74 M images/s instead of 1.34 M;
    # tools/stream_wait_probe.py.)
    a = copy.copy(args)
    a.mode, a.steps, a.warmup, a.no_extras, a.cpu_images = 'P', 200, 60, True, 6
    try:
        r = run_single(dict(env, args=a, all_cores=None))
        out['mode_p'] = {'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'], 'steps': 200,
                         'parity_ok': r.get('parity_ok'), 'parity_checked_images': r.get('parity_checked_images'),
                         'note': 'the headline batch decoded with utils/postprocess.py semantics (the variant whose oracle is '
                                 'pinned by outputs of the reference itself); PAFs rendered in its (y, x) channel order'}
        failed = failed or r.get('parity_ok') is False
    except Exception as exc:
        out['mode_p'] = {'error': repr(exc)}
        failed = True
    # (warm-up: long enough for every context to have recorded the launch sequences of the buffers it meets)
    for name, steps, warm, cpu_images in (('b256_656', 40, 24, 2), ('ms4_b128_368', 40, 24, 4), ('stream_b1024_368', 20, 6, 4)):
        a = copy.copy(args)
        a.workload, a.steps, a.warmup, a.no_extras, a.cpu_images = name, steps, warm, True, cpu_images
        a.contexts, a.depth, a.pipeline = 0, 0, ''
        wl = dict(WORKLOADS[name])
        wl['scenes'] = 32          # distinct scenes, repeated to the batch (rendering 256 crowded scenes takes a minute of CPU)
        multiscale = wl['kind'] == 'multiscale'
        a.pipeline = 'rounds' if wl['per_step'] > 1 else 'fetch'
        a.contexts = 3 if multiscale or a.pipeline == 'rounds' else 4
        e = dict(env, args=a, wl=wl, rounds=(a.pipeline == 'rounds') and not multiscale, all_cores=None)
        try:
            r = run_multiscale(e) if multiscale else run_single(e)
        except Exception as exc:            # a leg that fails is reported, it does not take the headline down
            legs[name] = {'error': repr(exc)}
            failed = True
            continue
        roof = r['roofline']
        legs[name] = {'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'], 'steps': steps,
                      'batch_per_step': r['config']['batch_per_gpu'], 'contexts': r['config']['contexts_per_gpu'],
                      'parity_ok': r.get('parity_ok'), 'parity_checked_images': r.get('parity_checked_images'),
                      'roofline': {k: roof.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic',
                                                            'algorithmic_bytes_per_launch', 'wasted_traffic_ratio', 'dominant_kernel')},
                      'cpu_baseline': r.get('cpu_baseline')}
        for k, v in roof.items():           # the per-kernel entries (event-timed in the leg's own region)
            if isinstance(v, dict) and 'avg_launch_ms' in v:
                legs[name]['roofline'][k] = v
        if 'kernels' in roof:
            legs[name]['roofline']['kernels'] = roof['kernels']
        failed = failed or r.get('parity_ok') is False
    out['other_workloads'] = legs
    try:
        out['detect_single_image'] = detect_latency(env)
        failed = failed or out['detect_single_image'].get('parity_ok') is False
    except Exception as exc:
        out['detect_single_image'] = {'error': repr(exc)}
        failed = True
    return failed


def detect_latency(env):
    """One 368 x 368 image, four test scales, through predict_decode.detect() (predict.py:45-254)
    with a stand-in network that returns rendered scenes: latency per image of everything but
    the network, host arrays in and Python structures out, beside the CPU restatement of the
    same loop."""
    from poseestimation_amd import predict_decode as pdec, synth
    from oracle import decode_ref as ref
    H = W = 368
    outs = {}
    for s in SCALES:
        hm, pf = synth.scene(H, W, 5, 4242, paf_order='xy', scale=s)
        outs[hm.shape[0]] = (hm[None], pf[None])

    def network(batch):             # (1, h, w, 3) padded image -> the scene's outputs at that size
        return outs[batch.shape[1] // 8]

    img = np.random.RandomState(3).randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    pdec.detect(img, network)       # tables, buffers
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        cand, peaks, subset = pdec.detect(img, network)
    t_detect = (time.perf_counter() - t0) / n
    pairs = [outs[k] for k in sorted(outs)]
    pads = [(0, 0)] * len(pairs)
    t0 = time.perf_counter()
    for _ in range(n):
        pdec.decode_scales(pairs, pads, (H, W))
    t_decode = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    want_h, want_p = ref.upsample_average([p[0][0] for p in pairs], [p[1][0] for p in pairs], pads, (H, W))
    w_cand, w_peaks, w_subset = ref.decode(want_h, want_p, 0.1, 0.05, 'D')
    t_cpu = time.perf_counter() - t0
    ok = w_subset.shape == subset.shape and np.array_equal(w_subset[:, :18], subset[:, :18]) \
        and np.allclose(w_subset[:, 18:], subset[:, 18:], rtol=1e-5, atol=0) and np.array_equal(w_cand[:, :2], cand[:, :2])
    in_bytes = sum(p[0].nbytes + p[1].nbytes for p in pairs)
    return {'image': '368x368, 5 people, scales 0.5 / 1 / 1.5 / 2', 'detect_ms': t_detect * 1e3,
            'decode_scales_ms': t_decode * 1e3, 'cpu_restatement_ms': t_cpu * 1e3, 'host_input_bytes': in_bytes,
            'persons': int(len(subset)), 'parity_ok': bool(ok),
            'note': 'detect_ms: image resize + pad of four scales on the device (host arrays in and out), the stand-in '
                    'network, then decode_scales; decode_scales_ms: predict.py:63-254 alone -- network outputs uploaded '
                    'from host arrays, one enqueue, results fetched and rebuilt as the reference\'s Python structures; '
                    'cpu_restatement_ms: oracle/decode_ref.py upsample_average + decode of the same outputs, one thread'}


def preheat(env, some_decodes):
    """Keeps the GPU busy with decodes for PD_BENCH_PREHEAT_MS (default 200) before the warm-up steps.  With several
    ranks the decision to go on is taken on the slowest rank's clock, so that all ranks issue the same gathers."""
    want = float(os.environ.get('PD_BENCH_PREHEAT_MS', '200')) * 1e-3
    t0 = time.perf_counter()
    while want > 0:
        some_decodes()
        if max_over_ranks(env, time.perf_counter() - t0) >= want:
            break


def make_barrier(env):
    torch, dist = env['torch'], env['dist']

    def barrier():
        torch.cuda.synchronize()         # device-wide: covers the contexts' own streams
        if env['use_dist']:
            if os.environ.get('PD_BENCH_BARRIER') == 'allreduce':
                dist.all_reduce(env['token'])
            elif os.environ.get('PD_BENCH_BARRIER') != 'none':
                dist.barrier()
        torch.cuda.synchronize()
    return barrier


def max_over_ranks(env, dt):
    if not env['use_dist']:
        return dt
    torch, dist = env['torch'], env['dist']
    t = torch.tensor([dt], dtype=torch.float64, device=env['comm_device'])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def pmc_traffic(workload):
    """Counter-derived HBM bytes per launch, committed under profiles/ by tools/summarize_profile.py."""
    path = os.path.join(ROOT, 'profiles', 'pmc_%s.json' % workload)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


# ---------------------------------------------------------------------------------------------
# single-scale workloads: b64_368, b256_656, stream_b1024_368
# ---------------------------------------------------------------------------------------------
def run_single(env):
    args, wl, world, rank, torch = env['args'], env['wl'], env['world'], env['rank'], env['torch']
    use_dist, rounds = env['use_dist'], env['rounds']
    from poseestimation_amd import _ffi, sharding, synth
    from poseestimation_amd.predict_decode import DecodeEngine, Results

    B, H, W, people = wl['B'], wl['H'], wl['W'], wl['people']
    desc = wl['desc']
    if args.people:      # experiments only: a fixed number of people per image
        people = args.people
        desc += ' (people per image overridden: %d)' % people
    h, w = H // 8, W // 8
    n_scenes = min(B, wl.get('scenes', B))
    heat, paf = synth.batch(n_scenes, H, W, people, seed0=rank * B, paf_order='xy' if args.mode == 'D' else 'yx')
    if n_scenes < B:        # short legs: fewer distinct scenes, repeated to the batch
        reps = -(-B // n_scenes)
        heat, paf = np.concatenate([heat] * reps)[:B], np.concatenate([paf] * reps)[:B]
        desc += ' (%d distinct scenes repeated to the batch)' % n_scenes
    # Distinct input buffers, used in turn, so that no step finds its inputs in the caches left
    # by the step before (together they exceed the 256 MiB Infinity Cache at 368^2): buffer k
    # holds the batch rotated by k images, i.e. image b of buffer k is image (b + k) % B.
    GATHER_DEPTH = int(os.environ.get('PD_BENCH_GATHER_DEPTH', '2'))   # decodes per context between two gathers
    n_ctx = args.contexts
    per_step = wl['per_step']
    if rounds:
        # one buffer per slice of a round; of two rounds where one round's inputs would fit the
        # 256 MiB Infinity Cache (the inputs of a step must come from HBM, as a network's outputs would)
        NBUF = n_ctx * GATHER_DEPTH
        two_rounds = per_step > 1 or NBUF * B * h * w * 57 * 4 < (320 << 20)
        if two_rounds:
            NBUF *= 2
    else:
        NBUF = 12 if B * h * w * 57 * 4 < (64 << 20) else 4
        NBUF = graph_friendly_buffers(NBUF, n_ctx)
    d_heat = [torch.from_numpy(np.roll(heat, -k, axis=0)).cuda() for k in range(NBUF)]
    d_paf = [torch.from_numpy(np.roll(paf, -k, axis=0)).cuda() for k in range(NBUF)]
    torch.cuda.synchronize()

    # several contexts (one HIP stream each) used round-robin: while the skeletons of
    # batch i are copied out and gathered, the kernels of the next batches already run,
    # and the short kernels of different batches overlap on the chip
    engs = [DecodeEngine(max_batch=B, max_h=H, max_w=W, mode=args.mode, device_id=env['local_rank'],
                         max_peaks_per_part=128, max_persons=128, max_pairs_per_image=40000)
            for _ in range(n_ctx)]
    # fetch pipeline: a second context on every stream, used in turn with the first, so that a
    # stream already holds the next decode while the host fetches the results of the last
    depth = 1
    if not rounds:
        # (decodes of large maps take a millisecond and more: the host's turn-around is nothing beside it)
        depth = args.depth if args.depth > 0 else (2 if h * w <= 4096 else 1)
        for d in range(1, depth):
            engs += [DecodeEngine(max_batch=B, max_h=H, max_w=W, mode=args.mode, device_id=env['local_rank'],
                                  max_peaks_per_part=128, max_persons=128, max_pairs_per_image=40000,
                                  share_stream_with=engs[i]) for i in range(n_ctx)]
    eng = engs[0]
    gather_cap = 1 << 18      # per rank and decode: header, count tables, peaks, person rows
    # (about 0.45 KB per person and image: 18 peaks of 16 bytes, a person row of 160)
    pmax = people if isinstance(people, int) else max(people)
    while gather_cap < B * 4096 * ((pmax + 7) // 8):
        gather_cap <<= 1

    gather = gather1 = None
    if rounds:
        gather = sharding.SkeletonGather([e.ctx for e in engs], gather_cap, depth=GATHER_DEPTH, local=not use_dist)
        gather1 = sharding.SkeletonGather([engs[0].ctx], gather_cap, depth=GATHER_DEPTH, local=True)
        for e in engs:      # everything a rank produces leaves through the gather
            e.ctx.set_host_results(False)

    host_t = [0.0, 0.0, 0.0, 0.0, 0, 0.0, 0]   # finish, submit, flush, start, rounds, longest start, its round
    counter = [0]                      # decodes issued so far: picks the input buffer

    def run(ndecodes, pool=None):
        """ndecodes batches decoded and their skeletons brought to the host (fetch: the packed
        results of every batch; rounds: peaks and person rows of a round of decodes in one
        copy, or -- N > 1 -- gathered to rank 0 over RCCL).  Returns (results, buffer) of the
        last decode."""
        pool_arg = pool
        if gather is None:
            pool = pool or engs
            n = len(pool)
            nbuf = graph_friendly_buffers(NBUF, n)
            pending, last = [], None
            for i in range(ndecodes):
                # context and input buffer both follow the count of decodes issued since the start,
                # so a context meets the same few buffers in every call of run() and replays the
                # launch sequences it has recorded for them, whatever the lengths of earlier runs
                e = pool[counter[0] % n]
                if len(pending) == n:
                    pe, pk = pending.pop(0)
                    last = (pe.ctx.fetch(), pk)     # waits for that stream; results now on the host
                k = counter[0] % nbuf
                counter[0] += 1
                e.enqueue_device(d_heat[k].data_ptr(), d_paf[k].data_ptr(), B, h, w, (H, W))
                pending.append((e, k))
            while pending:
                pe, pk = pending.pop(0)
                last = (pe.ctx.fetch(), pk)
            return last
        # rounds of GATHER_DEPTH decodes per context; the gather of round r (set r % 2 of the
        # send buffers) is in flight while round r + 1 decodes; the host never waits for a decode
        g, pool = gather, engs
        if pool_arg is not None:
            g, pool = gather1, pool_arg
        n = len(pool)
        last, r, done = None, 0, 0
        pending = [None, None]
        while done < ndecodes:
            s = r % 2
            ta = time.perf_counter()
            if pending[s] is not None:
                rows = g.finish(s)
                if rows is not None:
                    last = (rows, pending[s])
            tb = time.perf_counter()
            k = min(n * GATHER_DEPTH, ndecodes - done)
            bufs = []
            # the contexts' launcher threads issue the launches (or graph replays) in parallel
            for j in range(k):
                kb = (s * n * GATHER_DEPTH + j) % NBUF if two_rounds else j % NBUF
                bufs.append(kb)
                pool[j % n].ctx.submit(d_heat[kb].data_ptr(), d_paf[kb].data_ptr(), B, h, w, H, W, 0.1, 0.05,
                                       g.slice_ptr(s, j), g.capacity)
            tc = time.perf_counter()
            for e in pool:
                e.ctx.flush()
            td = time.perf_counter()
            if not g.on_gpu:        # (gloo rehearsal: the gather reads the send buffers from the host side)
                for e in pool:
                    e.ctx.sync()
            g.start(s)
            te = time.perf_counter()
            host_t[0] += tb - ta
            host_t[1] += tc - tb
            host_t[2] += td - tc
            host_t[3] += te - td
            if te - td > host_t[5]:
                host_t[5], host_t[6] = te - td, r
            host_t[4] += 1
            pending[s] = bufs
            done += k
            r += 1
        for s in (r % 2, (r + 1) % 2):
            if pending[s] is not None:
                rows = g.finish(s)
                if rows is not None:
                    last = (rows, pending[s])
        return last

    barrier = make_barrier(env)
    # Timed region: only one stage carries an event pair (one pair per decode); timing every
    # kernel would put two dozen event records between the kernels of every decode and slow
    # the job down by a third.  The full per-stage breakdown comes from a separate, untimed
    # pass afterwards.
    ROOF_STAGE = 'filter'
    # Python's cyclic collector walks every object the imports left behind (torch, numpy) at its
    # first full pass: a 35 ms stop some 650 steps into the run.  Those objects are parked in
    # the permanent generation, as a long-running server would do after start-up.
    import gc
    gc.collect()
    gc.freeze()
    for e in engs:
        e.ctx.profile(stages=[] if args.no_stage_timing else [ROOF_STAGE])
    # every context reaches its recorded launch sequences (third identical call per input
    # buffer and skeleton slice); with a process group, the first gathers also set up RCCL's
    # channels and the launcher threads
    run(3 * max(NBUF, len(engs) * GATHER_DEPTH * 2) if gather is None else 8 * len(engs) * GATHER_DEPTH)
    if use_dist:
        # One of the first twenty or so gathers of a process group holds the issuing thread for
        # 4-5 ms, once (measured with one rank: it fell into a 20-step region behind 9 rounds of
        # warm-up and into the next region behind 11; no later round did it again).  The warm-up
        # goes on until 24 rounds in a row have issued their gather in under a millisecond.
        quiet, total = 0, 0
        while quiet < 24 and total < 400:
            host_t[:] = [0.0, 0.0, 0.0, 0.0, 0, 0.0, 0]
            run(4 * len(engs) * GATHER_DEPTH)
            # (the ranks must agree on the number of gathers: the slowest rank's figure decides)
            quiet = quiet + 4 if max_over_ranks(env, host_t[5]) < 1e-3 else 0
            total += 4
    # The chip's clocks follow its load with a delay: behind a few milliseconds of warm-up a timed region of a millisecond
    # ran 5 to 20 % slower than the same region a moment later (regions of 20 steps one after the other in one process:
    # 0.94-1.18 M images/s for the first, 1.20-1.25 M for every later one).  So the chip is kept busy for a fifth of a
    # second before the warm-up steps asked for; PD_BENCH_PREHEAT_MS=0 leaves it out.
    preheat(env, lambda: run(len(engs) * (GATHER_DEPTH if gather is not None else 1) * 2))
    run(args.warmup * per_step)
    for e in engs:
        e.ctx.profile_read()
    barrier()
    host_t[:] = [0.0, 0.0, 0.0, 0.0, 0, 0.0, 0]
    t0 = time.perf_counter()
    last = run(args.steps * per_step)
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get('PD_BENCH_TIMERS') == '1' and host_t[4]:
        sys.stderr.write('host us per round: finish %.1f submit %.1f flush %.1f start %.1f (rounds %d)\n'
                         % tuple([x / host_t[4] * 1e6 for x in host_t[:4]] + [host_t[4]]))
    if os.environ.get('PD_BENCH_REGIONS'):
        # diagnostic: further timed regions of the given step counts, each bracketed like the
        # first, to see how the length of a region changes the rate it reports
        for k in [int(x) for x in os.environ['PD_BENCH_REGIONS'].split(',')]:
            barrier()
            host_t[:] = [0.0, 0.0, 0.0, 0.0, 0, 0.0, 0]
            ta = time.perf_counter()
            run(k * per_step)
            tc = time.perf_counter() - ta
            barrier()
            tb = time.perf_counter() - ta
            sys.stderr.write('region of %d steps: %.1f us per step, %.0f images/s (run %.0f us; host finish %.0f submit %.0f flush %.0f start %.0f us, longest %.0f us in round %d)\n'
                             % (k, tb / k * 1e6, B * per_step * k / tb, tc * 1e6, host_t[0] * 1e6, host_t[1] * 1e6, host_t[2] * 1e6, host_t[3] * 1e6,
                                host_t[5] * 1e6, host_t[6]))
    roof_ms, roof_n = 0.0, 0
    for e in engs:
        v = e.ctx.profile_read()[ROOF_STAGE]
        roof_ms += v[0]
        roof_n += v[1]
    dt = max_over_ranks(env, dt)

    # ---- untimed extra passes (rank 0 of a single-GPU run only) ----
    extras = rank == 0 and not use_dist and not args.no_extras
    prof, alone, dense_ms, host_rate = {}, None, None, None
    if extras:
        # the same event pair with one context decoding alone: the kernel's own duration, which is
        # what a kernel trace (rocprofv3 serialises the kernels of all streams) reports
        run(min(100, args.steps * per_step), pool=engs[:1])
        alone = engs[0].ctx.profile_read()[ROOF_STAGE]
        stage_decodes = max(10, min(50, args.steps * per_step))
        for e in engs:
            e.ctx.profile(True)
        run(3 * NBUF if gather is None else 8 * len(engs) * GATHER_DEPTH)
        for e in engs:
            e.ctx.profile_read()
        run(stage_decodes)
        for e in engs:
            for k, v in e.ctx.profile_read().items():
                a = prof.get(k, (0.0, 0))
                prof[k] = (a[0] + v[0], a[1] + v[1])
            e.ctx.profile(False)
        # the exact float64 kernel over every pixel (PD_IMPL_STREAM): the kernel that does the
        # work the byte formula of the peak stage prices (upsample, Gaussian, NMS on every pixel)
        dctx = _ffi.Context(device_id=env['local_rank'], max_batch=B, max_h=H, max_w=W, max_peaks_per_part=128,
                            max_persons=128, max_pairs_per_image=40000)
        dctx.set_mode(_ffi.MODE_D if args.mode == 'D' else _ffi.MODE_P)
        dctx.set_impl(_ffi.IMPL_STREAM)
        dctx.set_graph_replay(False)
        dctx.profile(stages=['smooth_nms'])
        for i in range(6):
            dctx.set_lowres(d_heat[0].data_ptr(), d_paf[0].data_ptr(), B, h, w, H, W, _ffi.MEM_DEVICE)
            dctx.peaks(0.1)
            if i == 1:
                dctx.profile_read()
        v = dctx.profile_read()['smooth_nms']
        dense_ms = v[0] / max(v[1], 1)
        dctx.close()
        # inputs handed over as host arrays (pinned): the reference's own boundary, predict.py:61
        if gather is None:
            ph = torch.from_numpy(heat).pin_memory()
            pp = torch.from_numpy(paf).pin_memory()
            nh = max(8, min(200, args.steps))
            pend = []

            def host_pass(count):
                for i in range(count):
                    e = engs[i % len(engs)]
                    if len(pend) == len(engs):
                        pend.pop(0).ctx.fetch()
                    e.enqueue(ph.numpy(), pp.numpy(), (H, W))
                    pend.append(e)
                while pend:
                    pend.pop(0).ctx.fetch()
            host_pass(2 * len(engs))
            torch.cuda.synchronize()
            th = time.perf_counter()
            host_pass(nh)
            torch.cuda.synchronize()
            host_rate = nh * B / (time.perf_counter() - th)

    out = None
    if rank == 0:
        images = world * B * per_step * args.steps
        n, N = h * w, H * W
        value = images / dt
        step_s = dt / args.steps
        rows, bufs = last
        if gather is None:
            res0, buf0 = rows, bufs
        else:
            allres = sharding.SkeletonGather.results(rows)
            res0, buf0 = allres[0], bufs[0]
        # Bytes one decode has to move for this algorithm: the network output read once, the
        # packed results written once.  (The per-image byte formula of a materialising pipeline -- x8 maps
        # written and read back -- prices an implementation that forms them; this path never forms them -- see `dense_path`.)
        result_bytes = int(len(res0['raw'])) if 'raw' in res0 else int(res0['peaks'].nbytes + res0['persons'].nbytes)
        must_bytes = B * n * 57 * 4.0 + result_bytes
        decode_s = step_s / per_step
        pmc = pmc_traffic(args.workload)
        # (the counters were read on decodes of pmc['batch'] images; other chunk sizes scale them)
        step_traffic = pmc.get('step_hbm_bytes') * (B / float(pmc.get('batch', B))) if pmc else None
        roofline = {
            'bound': 'hbm', 'kernel': 'whole decode of one batch (all kernels of the chain, %d contexts overlapped)' % len(engs),
            'achieved': must_bytes / decode_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': must_bytes / decode_s / 1e9 / HBM_PEAK_GBS,
            'traffic': step_traffic,
            'algorithmic_bytes_per_launch': must_bytes,
            'bytes_counted': 'network output of the batch read once (B*h*w*57*4) + packed results written once',
            'avg_launch_ms': decode_s * 1e3,
            'note': 'the fused path is bound by launch latency and vector-ALU issue, not by HBM: its kernels '
                    'visit only the few percent of the full-resolution map that can hold a peak',
        }
        if step_traffic:
            roofline['wasted_traffic_ratio'] = step_traffic / must_bytes
        if roof_n:
            f_s = roof_ms / roof_n * 1e-3
            # the kernel of the filter stage: maps of up to 4096 pixels take the fused form
            # (active cells + filter of a plane in one launch), larger ones the cell filter alone
            fused_small = n <= 4096 and os.environ.get('PD_SPLIT_CELLS', '0') in ('', '0')
            roof_kernel = 'k_cells_filter' if fused_small else 'k_cell_filter'
            kt = None
            if pmc and pmc.get('kernels'):
                for kname, kv in pmc['kernels'].items():
                    if kname.split('(')[0] == roof_kernel:
                        kt = kv['hbm_bytes']
            roofline[roof_kernel] = {
                'avg_launch_ms': f_s * 1e3, 'launches': roof_n,
                'measured': 'HIP event pair around the kernel in every decode of the timed region',
                'traffic': kt, 'achieved': (kt / f_s / 1e9) if kt else None,
                'frac': (kt / f_s / 1e9 / HBM_PEAK_GBS) if kt else None,
                'bound': 'valu issue + latency (profiles/*sq_counters*)'}
            if alone and alone[1]:
                roofline[roof_kernel]['one_context_avg_launch_ms'] = alone[0] / alone[1]
        if dense_ms:
            dense_bytes = B * (72.0 * n + 288.0 * N)
            peak_stage_ms = sum(prof[k][0] / prof[k][1] for k in ('planar', 'cells', 'filter', 'verify', 'smooth_nms')
                                if k in prof and prof[k][1])
            roofline['dense_path'] = {
                'kernel': 'k_smooth_nms_stream (PD_IMPL_STREAM: exact float64 kernel over every pixel)',
                'algorithmic_bytes_per_launch': dense_bytes,
                'bytes_counted': 'x8 upsample of 18 channels read + write, Gaussian read + write, NMS read: B*(72*n + 288*N)',
                'avg_launch_ms': dense_ms, 'achieved': dense_bytes / (dense_ms * 1e-3) / 1e9,
                'frac': dense_bytes / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'sparse_peak_stage_ms': peak_stage_ms,
                'dense_equivalent_speedup': (dense_bytes / (HBM_PEAK_GBS * 1e9) * 1e3) / peak_stage_ms if peak_stage_ms else None,
                'note': 'dense_equivalent_speedup = time a materialising pass would need at HBM peak / time of the '
                        'sparse peak stage (layout copy + cell scan + cell filter + verification)'}
        out = {
            'metric': 'decoded images/sec (heatmap+PAF->skeletons) at %dx%d' % (H, W),
            'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': step_s * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': desc, 'name': args.workload, 'batch_per_gpu': B * per_step, 'images_per_decode': B, 'mode': args.mode, 'preheat_ms': float(os.environ.get('PD_BENCH_PREHEAT_MS', '200')),
                       'contexts_per_gpu': len(engs), 'streams_per_gpu': n_ctx, 'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'default'),
                       'people_per_image': list(people) if isinstance(people, tuple) else people,
                       'inputs': 'device-resident network outputs, %d distinct batches of %.1f MB used in turn '
                                 '(the host-to-device copy of predict.py:61 is outside the timed region; '
                                 'see host_inputs_images_per_s)' % (NBUF, B * n * 57 * 4 / 1e6),
                       'outputs': 'full packed results (peaks, connection lists, person rows) stored to pinned host memory by the pack kernel of every decode'
                                  if gather is None else
                                  'peaks and person rows copied to the host once per round of %d decodes' % gather.n
                                  if not use_dist else
                                  'peaks and person rows of every rank gathered to rank 0 (RCCL) and copied to its host, '
                                  'once per round of %d decodes per rank' % gather.n},
            'roofline': roofline,
        }
        if prof:
            out['stage_ms_per_launch_untimed_pass'] = {k: v[0] / v[1] for k, v in prof.items() if v[1]}
        if host_rate:
            out['host_inputs_images_per_s'] = host_rate
            out['host_inputs_note'] = 'same decode with the network outputs in pinned host memory (PD_MEM_HOST): ' \
                                      'the copy over PCIe is inside this rate'
        out['persons_last_step'] = int(res0['person_count'].sum())
        if gather is None:
            out['filter_stats_last_step'] = eng.ctx.filter_stats()

        if not use_dist and not args.no_cpu:
            from oracle import decode_ref as ref
            ncpu = min(args.cpu_images, B)
            r = Results(res0, eng.mode)
            tc = time.perf_counter()
            ok = True
            for b in range(ncpu):
                src = (b + buf0) % B
                hf = ref.resize_cubic(heat[src], fx=8, fy=8).astype(np.float64)
                pf = ref.resize_cubic(paf[src], fx=8, fy=8).astype(np.float64)
                peaks = ref.get_peaks(hf, 0.1, args.mode, use_scipy=True)
                conns, special = ref.get_connections(peaks, pf, 0.05, args.mode)
                subset = ref.multi_person_parse(peaks, conns, special, args.mode)
                g = r.subset(b)
                ok = ok and g.shape == subset.shape and np.array_equal(g[:, :18], subset[:, :18]) \
                    and np.allclose(g[:, 18:], subset[:, 18:], rtol=1e-5, atol=0)
            tcpu = time.perf_counter() - tc
            out['cpu_baseline'] = {'value': ncpu / tcpu, 'unit': 'images/s', 'cores': 1, 'kind': 'port',
                                   'sample': '%d images of the same batch through oracle/decode_ref.py '
                                             '(numpy bicubic + scipy gaussian_filter + numpy decode), one thread; '
                                             'host has %d cpus' % (ncpu, os.cpu_count())}
            if env.get('all_cores'):
                out['cpu_baseline_all_cores'] = env['all_cores']
            out['parity_checked_images'] = ncpu
            out['parity_ok'] = bool(ok)
        if use_dist and not args.no_cpu and gather is not None and n_scenes == B:
            try:
                # N > 1: images of every rank, as they arrived on rank 0 through the gather, against the oracle on that
                # rank's inputs (rendered again here from their seeds); peaks and person rows are what the gather carries
                from oracle import decode_ref as ref
                r_all = Results(res0, eng.mode)
                ok, checked = True, 0
                for rk in range(world):
                    for b in sorted({0, B - 1}):
                        src = (b + buf0) % B
                        hm, pm = synth.batch(1, H, W, people, seed0=rk * B + src, paf_order='xy' if args.mode == 'D' else 'yx')
                        hf = ref.resize_cubic(hm[0], fx=8, fy=8).astype(np.float64)
                        pf = ref.resize_cubic(pm[0], fx=8, fy=8).astype(np.float64)
                        peaks = ref.get_peaks(hf, 0.1, args.mode, use_scipy=True)
                        conns, special = ref.get_connections(peaks, pf, 0.05, args.mode)
                        subset = ref.multi_person_parse(peaks, conns, special, args.mode)
                        g = r_all.subset(rk * B + b)
                        gp = r_all.all_peaks(rk * B + b)
                        ok = ok and g.shape == subset.shape and np.array_equal(g[:, :18], subset[:, :18]) \
                            and np.allclose(g[:, 18:], subset[:, 18:], rtol=1e-5, atol=0) \
                            and [len(x) for x in gp] == [len(x) for x in peaks] \
                            and all(int(a[0]) == int(c[0]) and int(a[1]) == int(c[1]) and float(a[2]) == float(c[2])
                                    for x, y in zip(gp, peaks) for a, c in zip(x, y))
                        checked += 1
                out['parity_checked_images'] = checked
                out['parity_ok'] = bool(ok)
                out['parity_note'] = 'two images of every rank from the last gathered round, as rank 0 received them'
            except Exception as exc:      # a check that cannot run must not take the measured line with it
                out['parity_ok'] = None
                out['parity_error'] = repr(exc)
    for e in engs:
        e.close()
    return out


# ---------------------------------------------------------------------------------------------
# multi-scale workload: ms4_b128_368  (predict.py:49-72, then the decode)
# ---------------------------------------------------------------------------------------------
def run_multiscale(env):
    args, wl, world, rank, torch = env['args'], env['wl'], env['world'], env['rank'], env['torch']
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine, Results

    B, H, W, people = wl['B'], wl['H'], wl['W'], wl['people']
    # 32 distinct scenes per rank, each rendered at the four test scales, repeated to the batch
    NSCENE = min(32, B, wl.get('scenes', B))
    rng = np.random.RandomState(100003 + rank)
    counts = [int(rng.randint(people[0], people[1] + 1)) for _ in range(NSCENE)]
    heats, pafs = [], []
    for s in SCALES:
        outs = [synth.scene(H, W, counts[i], 7000 + rank * NSCENE + i, paf_order='xy' if args.mode == 'D' else 'yx', scale=s)
                for i in range(NSCENE)]
        heats.append(np.stack([outs[b % NSCENE][0] for b in range(B)]))
        pafs.append(np.stack([outs[b % NSCENE][1] for b in range(B)]))
    sizes = [(a.shape[1], a.shape[2]) for a in heats]
    crops = [(8 * hs, 8 * ws) for hs, ws in sizes]          # 368 * scale is a multiple of 8: no padding to remove
    d_heats = [torch.from_numpy(a).cuda() for a in heats]
    d_pafs = [torch.from_numpy(a).cuda() for a in pafs]
    torch.cuda.synchronize()
    engs = [DecodeEngine(max_batch=B, max_h=H, max_w=W, mode=args.mode, device_id=env['local_rank'],
                         max_peaks_per_part=128, max_persons=128, max_pairs_per_image=40000, fullres=True)
            for _ in range(args.contexts)]
    hp = [t.data_ptr() for t in d_heats]
    pp = [t.data_ptr() for t in d_pafs]

    # N > 1: a rank's skeletons (peaks and person rows) leave through its contexts' skeleton
    # buffers and one gather per round of steps brings them to rank 0, as in the single-scale path
    gather = None
    if env['use_dist']:
        from poseestimation_amd import sharding
        gather_cap = 1 << 18
        while gather_cap < B * 4096:
            gather_cap <<= 1
        gather = sharding.SkeletonGather([e.ctx for e in engs], gather_cap, depth=1)
        for e in engs:
            e.ctx.set_host_results(False)

    stored = os.environ.get('PD_BENCH_MS_STORED') == '1'   # comparison runs: write the averaged maps out, decode those

    def decode(e):
        if stored:
            e.ctx.upsample_avg_device(hp, pp, sizes, crops, H, W, B)
            e.ctx.decode_fullres(0.1, 0.05)
        else:
            e.ctx.decode_multiscale(hp, pp, sizes, crops, H, W, B, 0.1, 0.05)

    def run(nsteps, pool=None):
        if gather is not None:
            n, r, done, last = len(engs), 0, 0, None
            busy = [False, False]
            while done < nsteps:
                s = r % 2
                if busy[s]:
                    rows = gather.finish(s)
                    if rows is not None:
                        last = rows
                k = min(n, nsteps - done)
                for j in range(k):
                    gather.point(s, j, j)
                    decode(engs[j])
                if not gather.on_gpu:
                    for e in engs[:k]:
                        e.ctx.sync()
                gather.start(s)
                busy[s] = True
                done += k
                r += 1
            for s in (r % 2, (r + 1) % 2):
                if busy[s]:
                    rows = gather.finish(s)
                    if rows is not None:
                        last = rows
            # (rank 0) the images of all ranks, in image order
            return None if last is None else sharding.SkeletonGather.results(last)[0]
        pool = pool or engs
        pending, last = [], None
        for i in range(nsteps):
            e = pool[i % len(pool)]
            if len(pending) == len(pool):
                last = pending.pop(0).ctx.fetch()
            decode(e)
            pending.append(e)
        while pending:
            last = pending.pop(0).ctx.fetch()
        return last

    barrier = make_barrier(env)
    import gc
    gc.collect()
    gc.freeze()
    # every kernel of a step carries an event pair: a step is seven launches of 0.02 - 0.4 ms, the
    # pairs cost nothing beside them
    TIMED = ['resize', 'planar', 'cells', 'filter', 'verify', 'smooth_nms', 'pairs', 'match', 'parse', 'pack']
    for e in engs:
        e.ctx.profile(stages=TIMED)
    run(2 * len(engs))
    if gather is not None:
        run(24)     # past the one slow gather among a process group's first twenty or so (see run_single)
    preheat(env, lambda: run(2 * len(engs)))      # (the chip's clocks: see run_single)
    run(min(args.warmup, 4))
    for e in engs:
        e.ctx.profile_read()
    barrier()
    t0 = time.perf_counter()
    last = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    stage = {}
    for e in engs:
        for k, v in e.ctx.profile_read().items():
            a = stage.get(k, (0.0, 0))
            stage[k] = (a[0] + v[0], a[1] + v[1])
    dt = max_over_ranks(env, dt)
    prof = {}
    if rank == 0 and gather is None and len(engs) > 1:
        # the same event pairs with one context decoding alone: a kernel's own duration (what a
        # kernel trace reports); with several contexts the kernels of different steps share the chip
        run(4, pool=engs[:1])
        engs[0].ctx.profile_read()
        run(10, pool=engs[:1])
        prof = engs[0].ctx.profile_read()

    out = None
    if rank == 0:
        value = world * B * args.steps / dt
        step_s = dt / args.steps
        n_sum = sum(hs * ws for hs, ws in sizes)
        N_sum = 64 * n_sum
        N = H * W
        pmc = pmc_traffic(args.workload)
        # Bytes a step has to move for this algorithm: the network outputs of every scale read once
        # (heatmaps without the background channel, which predict.py:91 never reads) and the packed
        # results written once.
        result_bytes = int(len(last['raw'])) if 'raw' in last else int(last['peaks'].nbytes + last['persons'].nbytes)
        must = B * 4.0 * n_sum * (18 + 38) + result_bytes
        traffic = pmc.get('step_hbm_bytes') if pmc else None
        roofline = {
            'bound': 'hbm', 'kernel': 'whole multi-scale decode of one batch (all kernels of the chain)',
            'achieved': must / step_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': must / step_s / 1e9 / HBM_PEAK_GBS, 'traffic': traffic,
            'algorithmic_bytes_per_launch': must,
            'bytes_counted': 'B * 4 * sum_s n_s * (18 + 38): the network outputs of the four scales read once '
                             '(no background heat channel) + packed results written once',
            'avg_launch_ms': step_s * 1e3,
            'note': 'nothing at full resolution is stored: the peak stage visits the cells that can hold a peak, the PAF '
                    'average is evaluated at the sample points of the candidate pairs; the path is bound by vector issue '
                    'and latency, not by HBM',
        }
        if traffic:
            roofline['wasted_traffic_ratio'] = traffic / must
        per_stage = {k: v[0] / v[1] for k, v in (prof or stage).items() if v[1]}
        # per-kernel figures: algorithmic bytes of each kernel of the chain / its event-timed duration
        kbytes = {
            'planar': ('k_ms_planar', B * 4.0 * n_sum * (19 + 18), 'heatmaps of the four scales read (19 channels), planar copies written (18)'),
            'cells': ('k_ms_cells', B * 4.0 * n_sum * 18, 'planar copies read once'),
            'filter': ('k_ms_cell_filter' if 'resize' not in per_stage else 'k_full_filter',
                       B * 4.0 * n_sum * 18 if 'resize' not in per_stage else B * 18.0 * N * 8,
                       'planar copies read once (the windows of the active cells overlap; re-reads come from the caches)'
                       if 'resize' not in per_stage else 'B * 18 * N * 8: the stored float64 averages read once'),
            'pairs': ('k_pair_scores_ms', B * 4.0 * n_sum * 38, 'PAF outputs of the four scales read once'),
            'resize': ('k_resize_tiles', B * (76.0 * n_sum + 19.0 * N * 8), 'heat outputs read, float64 averages written (stored-maps form)'),
        }
        kern = {}
        for st, (kname, nbytes, what) in kbytes.items():
            if st in per_stage:
                ms = per_stage[st]
                kt = None
                if pmc and pmc.get('kernels'):
                    for pk_name, kv in pmc['kernels'].items():
                        if pk_name.split('(')[0].split('<')[0] == kname:
                            kt = kv['hbm_bytes']
                kern[kname] = {'avg_launch_ms': ms, 'launches': (prof or stage)[st][1], 'algorithmic_bytes_per_launch': nbytes,
                               'bytes_counted': what, 'achieved': nbytes / (ms * 1e-3) / 1e9,
                               'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': kt,
                               'measured': 'HIP event pair around the kernel, one context decoding alone (ten steps behind the timed region)'
                                           if prof else 'HIP event pair around the kernel in every step of the timed region'}
        roofline['kernels'] = kern
        if per_stage:
            dom = max(per_stage, key=per_stage.get)
            roofline['dominant_kernel'] = kbytes[dom][0] if dom in kbytes else dom
        out = {
            'metric': 'decoded images/sec (heatmap+PAF->skeletons) at %dx%d' % (H, W),
            'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': wl['desc'], 'name': args.workload, 'batch_per_gpu': B, 'mode': args.mode,
                       'contexts_per_gpu': len(engs), 'scales': list(SCALES),
                       'inputs': 'device-resident network outputs of the four scales (%.0f MB per step; %d distinct '
                                 'scenes repeated to the batch)' % (B * 228.0 * n_sum / 1e6, NSCENE),
                       'outputs': 'full packed results copied to the host every step' if gather is None else
                                  'peaks and person rows of every rank gathered to rank 0 over RCCL, one gather per step'},
            'roofline': roofline,
            'stage_ms_per_step_timed_region': {k: v[0] / args.steps for k, v in stage.items() if v[1]},
        }
        if prof:
            out['stage_ms_per_launch_one_context'] = {k: v[0] / v[1] for k, v in prof.items() if v[1]}
        out['persons_last_step'] = int(last['person_count'].sum())
        if gather is None:
            out['filter_stats_last_step'] = engs[0].ctx.filter_stats()
        if not env['use_dist'] and not args.no_cpu:
            from oracle import decode_ref as ref
            ncpu = max(1, min(args.cpu_images // 4, 3))
            r = Results(last, engs[0].mode)
            ok = True
            tc = time.perf_counter()
            for b in range(ncpu):
                want_h, want_p = ref.upsample_average([a[b] for a in heats], [a[b] for a in pafs], [(0, 0)] * 4, (H, W))
                cand, peaks, subset = ref.decode(want_h, want_p, 0.1, 0.05, args.mode)
                g = r.subset(b)
                ok = ok and g.shape == subset.shape and np.array_equal(g[:, :18], subset[:, :18]) \
                    and np.allclose(g[:, 18:], subset[:, 18:], rtol=1e-5, atol=0) \
                    and np.array_equal(r.candidate(b)[:, :2], cand[:, :2])
            tcpu = time.perf_counter() - tc
            out['cpu_baseline'] = {'value': ncpu / tcpu, 'unit': 'images/s', 'cores': 1, 'kind': 'port',
                                   'sample': '%d images of the same batch through oracle/decode_ref.py (four-scale numpy '
                                             'bicubic + average, restated gaussian_filter, numpy decode), one thread; '
                                             'host has %d cpus' % (ncpu, os.cpu_count())}
            out['parity_checked_images'] = ncpu
            out['parity_ok'] = bool(ok)
        if env['use_dist'] and not args.no_cpu:
            try:
                # N > 1: one image of every rank, as it arrived on rank 0 through the gather, against the oracle on that
                # rank's inputs (rendered again here from their seeds)
                from oracle import decode_ref as ref
                r = Results(last, engs[0].mode)
                ok = True
                for rk in range(world):
                    rrng = np.random.RandomState(100003 + rk)
                    cnt = [int(rrng.randint(people[0], people[1] + 1)) for _ in range(NSCENE)]
                    b = (rk * 5) % NSCENE
                    outs = [synth.scene(H, W, cnt[b], 7000 + rk * NSCENE + b, paf_order='xy' if args.mode == 'D' else 'yx', scale=sc)
                            for sc in SCALES]
                    want_h, want_p = ref.upsample_average([o[0] for o in outs], [o[1] for o in outs], [(0, 0)] * 4, (H, W))
                    cand, peaks, subset = ref.decode(want_h, want_p, 0.1, 0.05, args.mode)
                    g = r.subset(rk * B + b)
                    ok = ok and g.shape == subset.shape and np.array_equal(g[:, :18], subset[:, :18]) \
                        and np.allclose(g[:, 18:], subset[:, 18:], rtol=1e-5, atol=0) \
                        and np.array_equal(r.candidate(rk * B + b)[:, :2], cand[:, :2])
                out['parity_checked_images'] = world
                out['parity_ok'] = bool(ok)
                out['parity_note'] = 'one image of every rank from the last gathered step, as rank 0 received it'
            except Exception as exc:
                out['parity_ok'] = None
                out['parity_error'] = repr(exc)
    for e in engs:
        e.close()
    return out


if __name__ == '__main__':
    main()
