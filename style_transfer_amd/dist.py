"""One-process-per-GPU tile farming over torch.distributed (RCCL on the GPU box, gloo in tests).

The in-process ``TileFarm`` drives several GPUs from one host process, which is what the
reference's pool of forked workers becomes on a single node.  Launchers such as
``python -m torch.distributed.run`` start one process per GPU instead; this module gives that
layout the same semantics.  Rank 0 is the master (it owns the image, the optimizer state and
the regularizers, like the reference's parent process); every rank, rank 0 included, is a tile
worker.  Per evaluation:

    rank 0: cut the tiles of the (virtually rolled) image        style_transfer.py:623-637
    scatter: tile t -> rank t mod world   (point-to-point sends, no collective reduction)
    all ranks: evaluate their tiles                               style_transfer.py:230-241
    gather: tile gradients and losses -> rank 0
    rank 0: stitch                                                style_transfer.py:639-643

Tiles never exchange data with each other, so the only traffic is 12 bytes per tile pixel in
each direction.  Targets (style Grams, content maps) are broadcast once per scale with
``broadcast_targets``.  The arithmetic is injected through three callables so that the
protocol can be exercised on CPU (gloo) in the unit tests.
"""

import numpy as np
import torch
import torch.distributed as dist


class DistributedTiles:
    """cut(rect) -> tensor[3,th,tw] on rank 0; evaluate(tile, start_yx, roll) -> (loss, grad
    tensor) on every rank; put(rect, grad) on rank 0.  All tensors live on ``device``."""

    def __init__(self, cut, evaluate, put, device, group=None):
        self.cut, self.evaluate, self.put = cut, evaluate, put
        self.device = torch.device(device)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def _sync(self):
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)

    def eval_sc_grad(self, rects, roll):
        """rects: the tile grid [(y0,y1,x0,x1)] (identical on all ranks); roll: (x, y) pixel
        shift, significant on rank 0.  Returns the summed loss on rank 0 (None elsewhere)."""
        header = torch.zeros(2, dtype=torch.int64, device=self.device)
        if self.rank == 0:
            header[0], header[1] = int(roll[0]), int(roll[1])
        dist.broadcast(header, 0, group=self.group)
        roll = (int(header[0]), int(header[1]))
        total = 0.0
        for base in range(0, len(rects), self.world):
            batch = rects[base:base + self.world]
            mine = batch[self.rank] if self.rank < len(batch) else None
            # ---- scatter: rank 0 sends tile r to rank r
            tile = None
            if self.rank == 0:
                sends = []
                for r, rect in enumerate(batch):
                    t = self.cut(rect, roll)
                    if r == 0:
                        tile = t
                    else:
                        sends.append(dist.isend(t.contiguous(), r, group=self.group))
                self._sync()
                for s in sends:
                    s.wait()
            elif mine is not None:
                tile = torch.empty((3, mine[1] - mine[0], mine[3] - mine[2]), dtype=torch.float32,
                                   device=self.device)
                dist.recv(tile, 0, group=self.group)
            self._sync()
            # ---- evaluate
            loss, grad = (0.0, None)
            if mine is not None:
                loss, grad = self.evaluate(tile, (mine[0], mine[2]), roll)
            # ---- gather
            if self.rank == 0:
                if mine is not None:
                    self.put(mine, grad, roll)
                    total += loss
                for r, rect in enumerate(batch):
                    if r == 0:
                        continue
                    buf = torch.empty((3, rect[1] - rect[0], rect[3] - rect[2]),
                                      dtype=torch.float32, device=self.device)
                    dist.recv(buf, r, group=self.group)
                    lbuf = torch.zeros(1, dtype=torch.float64, device=self.device)
                    dist.recv(lbuf, r, group=self.group)
                    self._sync()
                    self.put(rect, buf, roll)
                    total += float(lbuf[0])
            elif mine is not None:
                self._sync()
                dist.send(grad.contiguous(), 0, group=self.group)
                dist.send(torch.tensor([loss], dtype=torch.float64, device=self.device), 0,
                          group=self.group)
        return total if self.rank == 0 else None


def broadcast_targets(contents, styles, device, group=None):
    """Broadcasts rank 0's targets (lists of {layer: ndarray}) to every rank."""
    rank = dist.get_rank(group)
    meta = [None]
    if rank == 0:
        meta[0] = ([{k: v.shape for k, v in c.items()} for c in contents],
                   [{k: v.shape for k, v in s.items()} for s in styles])
    dist.broadcast_object_list(meta, 0, group=group)
    cshapes, sshapes = meta[0]
    out = []
    for shapes, src in ((cshapes, contents), (sshapes, styles)):
        items = []
        for i, shape_map in enumerate(shapes):
            d = {}
            for layer, shape in shape_map.items():
                if rank == 0:
                    t = torch.from_numpy(np.ascontiguousarray(src[i][layer], np.float32)).to(device)
                else:
                    t = torch.empty(shape, dtype=torch.float32, device=device)
                dist.broadcast(t, 0, group=group)
                d[layer] = t.cpu().numpy()
            items.append(d)
        out.append(items)
    return out[0], out[1]
