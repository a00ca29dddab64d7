"""Image-sharded decode across the GPUs of one node.

Images are independent (the peak-id counter restarts per image,
utils/postprocess.py:36-37), so a batch is cut into contiguous blocks, one per rank,
each decoded on its own GPU and HIP stream with no data-path collective.  The only
exchange is the gather of the variable-length skeleton lists to rank 0, done with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU).
Rank-major order of the gathered blocks is the original image order.
"""
import numpy as np

from ._ffi import CONN_DTYPE, PEAK_DTYPE

_MAGIC = 0x50444B31  # "PDK1"


def shard_range(n_images, world_size, rank):
    """Contiguous block [lo, hi) of rank ``rank``: ceil(n / world) images per rank."""
    per = -(-n_images // world_size)
    lo = min(n_images, rank * per)
    return lo, min(n_images, lo + per)


def pack_blob(res):
    """Serialise the packed results of one rank (the dict of ``Context.fetch``) into a
    flat uint8 array: header, count tables, then peaks / connections / persons."""
    B = int(res['B'])
    head = np.array([_MAGIC, B, len(res['peaks']), len(res['conns']), len(res['persons'])], np.int64)
    parts = [head.view(np.uint8),
             np.ascontiguousarray(res['part_count'], np.int32).view(np.uint8).ravel(),
             np.ascontiguousarray(res['limb_count'], np.int32).view(np.uint8).ravel(),
             np.ascontiguousarray(res['person_count'], np.int32).view(np.uint8).ravel(),
             np.ascontiguousarray(res['status'], np.int32).view(np.uint8).ravel(),
             np.ascontiguousarray(res['peaks']).view(np.uint8).ravel(),
             np.ascontiguousarray(res['conns']).view(np.uint8).ravel(),
             np.ascontiguousarray(res['persons'], np.float64).view(np.uint8).ravel()]
    return np.concatenate(parts)


def blob_nbytes(blob):
    head = np.frombuffer(blob[:40].tobytes(), np.int64)
    if head[0] != _MAGIC:
        raise ValueError('not a result blob')
    B, npk, ncn, nps = (int(v) for v in head[1:5])
    return 40 + B * (18 + 19 + 1 + 1) * 4 + npk * PEAK_DTYPE.itemsize + ncn * CONN_DTYPE.itemsize + nps * 160


def unpack_blob(blob):
    blob = np.ascontiguousarray(blob, np.uint8)
    head = np.frombuffer(blob[:40].tobytes(), np.int64)
    if head[0] != _MAGIC:
        raise ValueError('not a result blob')
    B, npk, ncn, nps = (int(v) for v in head[1:5])
    at = [40]

    def take(nbytes, dtype, shape):
        a = np.frombuffer(blob[at[0]:at[0] + nbytes].tobytes(), dtype).reshape(shape)
        at[0] += nbytes
        return a
    return {
        'B': B,
        'part_count': take(B * 18 * 4, np.int32, (B, 18)),
        'limb_count': take(B * 19 * 4, np.int32, (B, 19)),
        'person_count': take(B * 4, np.int32, (B,)),
        'status': take(B * 4, np.int32, (B,)),
        'peaks': take(npk * PEAK_DTYPE.itemsize, PEAK_DTYPE, (npk,)),
        'conns': take(ncn * CONN_DTYPE.itemsize, CONN_DTYPE, (ncn,)),
        'persons': take(nps * 160, np.float64, (nps, 20)),
    }


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
