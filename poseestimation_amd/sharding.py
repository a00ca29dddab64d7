"""Image-sharded decode across the GPUs of one node.

Images are independent (the peak-id counter restarts per image,
utils/postprocess.py:36-37), so a batch is cut into contiguous blocks, one per rank,
each decoded on its own GPU and HIP stream with no data-path collective.  The only
exchange is the gather of the variable-length skeleton lists to rank 0, done with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU).
Rank-major order of the gathered blocks is the original image order.
"""
import numpy as np

from ._ffi import CONN_DTYPE, PD_E_INDEX, PEAK_DTYPE, DecodeOverflow, res_layout, unpack_results


def shard_range(n_images, world_size, rank):
    """Contiguous block [lo, hi) of rank ``rank``: ceil(n / world) images per rank."""
    per = -(-n_images // world_size)
    lo = min(n_images, rank * per)
    return lo, min(n_images, lo + per)


def pack_blob(res):
    """The packed results of one rank as a flat uint8 array in the library's own
    result-buffer layout (header, count tables, peaks, connections, persons).  A dict
    that came from ``Context.fetch`` already carries that image as ``res['raw']``."""
    if 'raw' in res:
        return res['raw']
    B = int(res['B'])
    off_part, off_limb, off_pers, off_stat, off_pay = res_layout(B)
    npk, ncn, nps = len(res['peaks']), len(res['conns']), len(res['persons'])
    raw = np.zeros(off_pay + npk * PEAK_DTYPE.itemsize + ncn * CONN_DTYPE.itemsize + nps * 160, np.uint8)
    raw[:64].view(np.int64)[:5] = [B, npk, ncn, nps, -1]
    raw[off_part:off_limb] = np.ascontiguousarray(res['part_count'], np.int32).view(np.uint8).ravel()
    raw[off_limb:off_pers] = np.ascontiguousarray(res['limb_count'], np.int32).view(np.uint8).ravel()
    raw[off_pers:off_stat] = np.ascontiguousarray(res['person_count'], np.int32).view(np.uint8).ravel()
    raw[off_stat:off_stat + 4 * B] = np.ascontiguousarray(res['status'], np.int32).view(np.uint8).ravel()
    p1 = off_pay + npk * PEAK_DTYPE.itemsize
    p2 = p1 + ncn * CONN_DTYPE.itemsize
    raw[off_pay:p1] = np.ascontiguousarray(res['peaks']).view(np.uint8).ravel()
    raw[p1:p2] = np.ascontiguousarray(res['conns']).view(np.uint8).ravel()
    raw[p2:] = np.ascontiguousarray(res['persons'], np.float64).view(np.uint8).ravel()
    return raw


def blob_nbytes(blob):
    head = np.ascontiguousarray(blob[:64]).view(np.int64)
    B, npk, ncn, nps = (int(v) for v in head[:4])
    if B < 0 or npk < 0 or ncn < 0 or nps < 0:
        raise ValueError('not a result blob')
    return res_layout(B)[4] + npk * PEAK_DTYPE.itemsize + ncn * CONN_DTYPE.itemsize + nps * 160


def unpack_blob(blob):
    blob = np.ascontiguousarray(blob, np.uint8)
    return unpack_results(blob[:blob_nbytes(blob)].copy())


def concat_results(parts):
    """Merge per-rank result dicts (rank-major) into one, restoring image order."""
    out = {'B': sum(p['B'] for p in parts)}
    for key in ('part_count', 'limb_count', 'person_count', 'status', 'peaks', 'conns', 'persons'):
        out[key] = np.concatenate([p[key] for p in parts], axis=0)
    return out


def gather_blobs(blob, capacity, group=None, dst=0):
    """One fixed-size gather of every rank's blob to ``dst``.  ``capacity`` is the
    padded per-rank size in bytes, the same on all ranks; a blob that does not fit
    raises on its rank.  Returns the list of blobs on ``dst`` and None elsewhere."""
    import torch
    import torch.distributed as dist
    n = int(blob.size)
    if n > capacity:
        raise ValueError('result blob of %d bytes exceeds the gather capacity %d' % (n, capacity))
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    on_gpu = dist.get_backend(group) == 'nccl'
    dev = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
    send = torch.zeros(capacity, dtype=torch.uint8, device=dev)
    send[:n] = torch.from_numpy(blob).to(dev)
    if rank == dst:
        recv = [torch.empty(capacity, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(send, gather_list=recv, dst=dst, group=group)
        out = []
        for t in recv:
            a = t.cpu().numpy()
            out.append(a[:blob_nbytes(a)])
        return out
    dist.gather(send, gather_list=None, dst=dst, group=group)
    return None


def gather_results(res, capacity, group=None, dst=0):
    """Gather every rank's packed results to ``dst`` and merge them in image order."""
    blobs = gather_blobs(pack_blob(res), capacity, group, dst)
    if blobs is None:
        return None
    return concat_results([unpack_blob(b) for b in blobs])


class SkeletonGather:
    """The exchange of a sharded decode: every rank's peaks and person rows (the
    library's skeleton buffer, see ``Context.set_skeleton_buffer``) are gathered to rank
    ``dst`` with one fixed-size ``dist.gather`` per *round* -- ``depth`` decodes on each
    of the rank's contexts -- straight from the device buffers the pack kernels wrote
    (backend "nccl" = RCCL; with "gloo" the buffers are staged through host memory, which
    is what the CPU tests run).

    There are ``sets`` send buffers (two: the decodes of round r + 1 run while the gather
    of round r is in flight), each with ``contexts * depth`` slices of ``capacity`` bytes.
    Per round: ``point(s, j, i)`` directs the next decode of context ``i`` to slice ``j``
    of set ``s`` and the caller enqueues that decode (or hands ``slice_ptr(s, j)`` to
    ``Context.submit`` and flushes the contexts); ``start(s)`` launches the gather
    behind the round's decodes (on the GPU the collective waits for the contexts' streams
    itself, the host does not block); ``finish(s)`` -- at the latest before set ``s`` is
    used again -- waits for it and returns, on ``dst``, a ``(world, slices, capacity)``
    uint8 array: rank-major, then in slice order.
    """

    def __init__(self, contexts, capacity, group=None, dst=0, device=None, sets=2, depth=1, local=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group, self.dst, self.capacity = group, dst, int(capacity)
        self.contexts = list(contexts)
        # local: a single process without a process group; the "gather" is then the copy of
        # the send set to pinned host memory, with the same rounds and the same result rows
        self.local = bool(local)
        if self.local:
            self.world, self.rank, self.dst, self.on_gpu = 1, 0, 0, True
        else:
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
            self.on_gpu = dist.get_backend(group) == 'nccl'
        dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        n = self.n = len(self.contexts) * int(depth)   # slices per set
        row = n * self.capacity
        self.send = [torch.zeros(row, dtype=torch.uint8, device=dev) for _ in range(sets)]
        self.stage = None if self.on_gpu else [torch.zeros(row, dtype=torch.uint8) for _ in range(sets)]
        self.recv = self.host = None
        if self.rank == self.dst:
            rdev = dev if self.on_gpu else torch.device('cpu')
            if not self.local:
                self.recv = [torch.zeros(self.world * row, dtype=torch.uint8, device=rdev) for _ in range(sets)]
            if self.on_gpu:
                self.host = [torch.zeros(self.world * row, dtype=torch.uint8).pin_memory() for _ in range(sets)]
        # the root's receive views, one list per set, made once (not per round)
        self.recv_views = None
        if self.recv is not None:
            self.recv_views = [list(r.view(self.world, -1).unbind(0)) for r in self.recv]
        self.work = [None] * sets
        self.streams = self.copy_stream = None
        if self.on_gpu:
            # the contexts' own HIP streams, as torch sees them
            self.streams = [torch.cuda.ExternalStream(int(c.stream), device=dev) for c in self.contexts]
            self.copy_stream = torch.cuda.Stream(device=dev)
            self.done = [torch.cuda.Event() for _ in range(sets)]

    def slice_ptr(self, s, j):
        """Device address of slice ``j`` of set ``s`` (for ``Context.submit``)."""
        return self.send[s].data_ptr() + j * self.capacity

    def point(self, s, j, i):
        self.contexts[i].set_skeleton_buffer(self.slice_ptr(s, j), self.capacity)

    def start(self, s):
        src = self.send[s]
        if self.local:
            for st in self.streams:
                self.copy_stream.wait_stream(st)   # the copy is ordered behind this round's decodes
            with self.torch.cuda.stream(self.copy_stream):
                self._to_host(src, self.host[s], self.n)
                self.done[s].record()
            self.work[s] = True
            return
        if self.on_gpu:
            cur = self.torch.cuda.current_stream()
            for st in self.streams:
                cur.wait_stream(st)      # the collective is ordered behind this round's decodes
        else:
            self.stage[s].copy_(src)     # the caller has waited for the contexts
            src = self.stage[s]
        gl = self.recv_views[s] if self.rank == self.dst else None
        self.work[s] = self.dist.gather(src, gather_list=gl, dst=self.dst, group=self.group, async_op=True)
        if self.on_gpu:
            # wait() orders a stream behind the collective without blocking the host; the copy to
            # pinned memory is queued right away on a stream of its own, so that finish() finds
            # it done (or under way) and waits for nothing else
            with self.torch.cuda.stream(self.copy_stream):
                self.work[s].wait()
                if self.rank == self.dst:
                    self._to_host(self.recv[s], self.host[s], self.world * self.n)
                self.done[s].record()

    def _to_host(self, src, dst, n_slices):
        """Queues, on the current torch stream, the copy of ``n_slices`` slices to pinned host memory.  Up to 8 MiB a
        plain copy (the copy engines; nothing of the chip is taken from the decodes).  Beyond that -- eight ranks'
        slices of a round -- of every slice only the bytes its header says are in use (``pd_copy_used_skeletons``: a
        slice is sized for the largest result and typically half full, and whole slices of eight ranks are 40 GB/s
        and more on the root's PCIe link; the kernel costs the root's decodes 1.4 %, measured with one rank).  What
        lies behind a slice's used part in ``dst`` is then left as it was.  PD_GATHER_PLAIN_COPY=1 / 0 in the
        environment forces one or the other."""
        import os
        forced = os.environ.get('PD_GATHER_PLAIN_COPY', '')
        plain = n_slices * self.capacity <= (8 << 20) if forced not in ('0', '1') else forced == '1'
        if plain or self.capacity % 16 or self.capacity < 64:
            dst.copy_(src, non_blocking=True)
            return
        from . import _ffi
        L = _ffi.lib()
        rc = L.pd_copy_used_skeletons(src.data_ptr(), dst.data_ptr(), n_slices, self.capacity,
                                      self.torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError('pd_copy_used_skeletons failed (%d): %s' % (rc, L.pd_last_error(None).decode()))

    def finish(self, s):
        work, self.work[s] = self.work[s], None
        if work is None:
            return None
        out = None
        if self.on_gpu:
            self.done[s].synchronize()   # this set's copy only: the next round's is queued behind it
            if self.rank == self.dst:
                out = self.host[s].numpy()
        else:
            work.wait()
            if self.rank == self.dst:
                out = self.recv[s].numpy()
        return None if out is None else out.reshape(self.world, self.n, self.capacity)

    @staticmethod
    def results(rows):
        """``finish``'s array -> one result dict per slice (= per step of the round), each
        with the images of all ranks in image order (connections are not part of the
        exchange: ``conns`` is empty and ``limb_count`` zero)."""
        out = []
        for i in range(rows.shape[1]):
            parts = []
            for r in range(rows.shape[0]):
                row = rows[r, i]
                need = int(np.ascontiguousarray(row[:64]).view(np.int64)[5])
                if need:
                    raise ValueError('rank %d: its skeleton blob needs %d bytes, the gather capacity is %d'
                                     % (r, need, row.size))
                part = unpack_blob(row)
                # a skeleton-only decode has no pd_sync_totals to report a failed image: the
                # header's first_bad_image and the status table carry it across the gather
                bad = int(np.ascontiguousarray(row[:64]).view(np.int64)[4])
                if bad >= 0:
                    st = int(part['status'][bad])
                    where = 'rank %d, step slice %d, image %d of its shard' % (r, i, bad)
                    if st == PD_E_INDEX:
                        raise IndexError('%s: a connection touches more than two persons '
                                         '(predict.py:199 IndexError)' % where)
                    raise DecodeOverflow('%s: a capacity of the decode context was exceeded (status %d)'
                                         % (where, st))
                parts.append(part)
            out.append(concat_results(parts))
        return out
