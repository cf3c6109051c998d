"""Multi-GPU frame rendering: image tiles across ranks, one gather of tiles to rank 0.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" on CPU for
tests).  The path shards with no data-path exchange: pixels are independent, the scene (a few
MB) is replicated, and rank r renders tiles r, r+world, ... of the 8x8 tile grid
(``hrt_render_tiles``).  The only collective is ONE ``gather`` of each rank's dense tile buffer
to rank 0; rank 0 then de-interleaves tiles into the row-major frame
(``hrt_assemble_frame``).  There is no reduction: ranks own disjoint pixels.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import TILE, assemble_frame, assemble_frame_host, tiles_owned, tiles_total


def padded_tiles_per_rank(w: int, h: int, world: int) -> int:
    """Tiles of rank 0 (the largest share); every rank's gather buffer is padded to this."""
    return tiles_owned(w, h, 0, world)


def render_frame_distributed(render_tiles: Callable, w: int, h: int, rank: int, world: int, device,
                             on_gpu: bool, stream_ptr: int = 0):
    """Render this rank's tiles and gather all tiles on rank 0.

    ``render_tiles(buffer_tensor)`` must fill ``buffer_tensor`` (float32, shape
    (padded_tiles, TILE*TILE, 3), on ``device``) with this rank's tiles in local order.
    Returns the (h, w, 3) frame tensor on rank 0 and ``None`` elsewhere.
    """
    import torch
    import torch.distributed as dist

    per_rank = padded_tiles_per_rank(w, h, world)
    mine = torch.zeros((per_rank, TILE * TILE, 3), dtype=torch.float32, device=device)
    render_tiles(mine)
    if world == 1:
        gathered = mine.unsqueeze(0)
    else:
        if dist.get_backend() == "gloo" and mine.is_cuda:  # gloo has no device gather: stage through host memory
            torch.cuda.current_stream().synchronize()
            host = mine.cpu()
            bufs = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, gather_list=bufs, dst=0)
            if rank != 0:
                return None
            gathered = torch.stack(bufs, dim=0).to(device)
        else:
            bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, gather_list=bufs, dst=0)
            if rank != 0:
                return None
            gathered = torch.stack(bufs, dim=0)
    if on_gpu:
        frame = torch.empty((h, w, 3), dtype=torch.float32, device=device)
        assemble_frame(gathered.data_ptr(), per_rank, w, h, world, frame.data_ptr(), stream_ptr)
        return frame
    return torch.from_numpy(assemble_frame_host(gathered.cpu().numpy(), w, h, world))


def tile_pixels(w: int, h: int, rank: int, world: int):
    """(local_slot, x0, y0) of every tile owned by ``rank`` -- the host statement of the kernel's mapping."""
    tx = (w + TILE - 1) // TILE
    out = []
    for slot, t in enumerate(range(rank, tiles_total(w, h), world)):
        out.append((slot, (t % tx) * TILE, (t // tx) * TILE))
    return out
