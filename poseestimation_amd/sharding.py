"""Image-sharded decode across the GPUs of one node.

Images are independent (the peak-id counter restarts per image,
utils/postprocess.py:36-37), so a batch is cut into contiguous blocks, one per rank,
each decoded on its own GPU and HIP stream with no data-path collective.  The only
exchange is the gather of the variable-length skeleton lists to rank 0, done with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU).
Rank-major order of the gathered blocks is the original image order.
"""
import numpy as np

from ._ffi import CONN_DTYPE, PEAK_DTYPE, res_layout, unpack_results


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
    """The per-step exchange of a sharded decode: every rank's peaks and person rows
    (the library's skeleton buffer, see ``Context.set_skeleton_buffer``) are gathered to
    rank ``dst`` with one fixed-size ``dist.gather`` per step, straight from the device
    buffer the pack kernel wrote (backend "nccl" = RCCL; with "gloo" the buffer is
    staged through host memory, which is what the CPU tests run).

    One slot per decode context.  ``start(slot)`` launches the gather of that context's
    buffer (its stream must have been waited for), ``finish(slot)`` waits for it and
    returns, on ``dst``, a ``(world, capacity)`` uint8 array of the ranks' blobs in rank
    order = image order; the slot's buffer may be overwritten by the next decode only
    after ``finish``.
    """

    def __init__(self, contexts, capacity, group=None, dst=0, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group, self.dst, self.capacity = group, dst, int(capacity)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.on_gpu = dist.get_backend(group) == 'nccl'
        dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        n = len(contexts)
        self.send = [torch.zeros(self.capacity, dtype=torch.uint8, device=dev) for _ in range(n)]
        for ctx, t in zip(contexts, self.send):
            ctx.set_skeleton_buffer(t.data_ptr(), self.capacity)
        self.stage = None if self.on_gpu else [torch.zeros(self.capacity, dtype=torch.uint8) for _ in range(n)]
        self.recv = self.host = None
        if self.rank == dst:
            rdev = dev if self.on_gpu else torch.device('cpu')
            self.recv = [torch.zeros(self.world * self.capacity, dtype=torch.uint8, device=rdev) for _ in range(n)]
            if self.on_gpu:
                self.host = [torch.zeros(self.world * self.capacity, dtype=torch.uint8).pin_memory() for _ in range(n)]
        self.work = [None] * n

    def start(self, slot):
        src = self.send[slot]
        if not self.on_gpu:
            self.stage[slot].copy_(src)
            src = self.stage[slot]
        gl = None
        if self.rank == self.dst:
            gl = list(self.recv[slot].view(self.world, self.capacity).unbind(0))
        self.work[slot] = self.dist.gather(src, gather_list=gl, dst=self.dst, group=self.group, async_op=True)

    def finish(self, slot):
        work, self.work[slot] = self.work[slot], None
        if work is None:
            return None
        work.wait()
        out = None
        if self.on_gpu:
            if self.rank == self.dst:
                self.host[slot].copy_(self.recv[slot], non_blocking=True)
            # wait() only orders the current stream behind the collective: the host has to
            # wait too before the send buffer is handed back to the decode stream
            self.torch.cuda.current_stream().synchronize()
            if self.rank == self.dst:
                out = self.host[slot].numpy()
        elif self.rank == self.dst:
            out = self.recv[slot].numpy()
        return None if out is None else out.reshape(self.world, self.capacity)

    @staticmethod
    def results(rows):
        """Blobs of all ranks -> one result dict in image order (connections are not part
        of the exchange: ``conns`` is empty and ``limb_count`` zero)."""
        parts = []
        for r, row in enumerate(rows):
            need = int(np.ascontiguousarray(row[:64]).view(np.int64)[5])
            if need:
                raise ValueError('rank %d: its skeleton blob needs %d bytes, the gather capacity is %d'
                                 % (r, need, row.size))
            parts.append(unpack_blob(row))
        return concat_results(parts)
