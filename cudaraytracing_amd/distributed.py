"""Image-space partition across GPUs and the one exchange step of a frame.

Pixels are independent (one lane per pixel, seed = global pixel index), so every
rank renders the 16-row strips it owns (strip s belongs to rank s % world) with the
scene replicated, and the only communication is one gather of the owned rows to
rank 0 (RCCL over xGMI when the tensors live on GPUs, gloo on CPU).  No reduction is
involved: the gathered image is bit-identical to a single-GPU render."""
import numpy as np
import torch
import torch.distributed as dist


def owned_rows(part, parts, height, strip_rows=16):
    """Row indices of the image that `part` of `parts` renders."""
    rows = np.arange(height)
    return rows[(rows // strip_rows) % parts == part]


_INDEX_CACHE = {}


def _row_indices(parts, height, strip_rows, device):
    """Index tensors of every rank's rows, built once per image shape and device."""
    key = (parts, height, strip_rows, str(device))
    if key not in _INDEX_CACHE:
        _INDEX_CACHE[key] = [torch.as_tensor(owned_rows(p, parts, height, strip_rows), device=device, dtype=torch.long)
                             for p in range(parts)]
    return _INDEX_CACHE[key]


_BUFFER_CACHE = {}


def _gather_buffers(local, parts, max_rows, is_dst):
    """Staging buffers of one image shape, allocated once: the packed rows of this rank and, on the destination rank, one
    receive buffer per rank and the assembled image."""
    key = (tuple(local.shape), local.dtype, str(local.device), parts, is_dst)
    if key not in _BUFFER_CACHE:
        packed = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        bufs = [torch.empty_like(packed) for _ in range(parts)] if is_dst else None
        out = torch.empty_like(local) if is_dst else None
        _BUFFER_CACHE[key] = (packed, bufs, out)
    return _BUFFER_CACHE[key]


def gather_rows(local, part, parts, strip_rows=16, dst=0, group=None):
    """Gather the owned rows of every rank's (H, W, C) tensor to rank `dst`.

    `local` holds valid data in the rows this rank owns.  Returns the assembled
    (H, W, C) tensor on `dst` and None elsewhere.  With parts == 1 returns `local`.
    The staging buffers and the assembled image are allocated once per image shape and
    reused: the returned tensor is overwritten by the next call with the same shape."""
    if parts == 1:
        return local
    H = local.shape[0]
    idx = _row_indices(parts, H, strip_rows, local.device)
    max_rows = max(int(i.numel()) for i in idx)
    packed, bufs, out = _gather_buffers(local, parts, max_rows, part == dst)
    mine = idx[part]
    torch.index_select(local, 0, mine, out=packed[: mine.numel()])
    if part == dst:
        dist.gather(packed, gather_list=bufs, dst=dst, group=group)
        for p in range(parts):
            out.index_copy_(0, idx[p], bufs[p][: idx[p].numel()])
        return out
    dist.gather(packed, gather_list=None, dst=dst, group=group)
    return None
