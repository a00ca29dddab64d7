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
