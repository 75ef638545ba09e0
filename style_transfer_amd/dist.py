"""One-process-per-GPU tile farming over torch.distributed (RCCL on the GPU box, gloo in tests).

The in-process ``TileFarm`` drives several GPUs from one host process, which is what the
reference's pool of forked workers becomes on a single node.  Launchers such as
``python -m torch.distributed.run`` start one process per GPU instead; this module gives that
layout the same semantics.  Rank 0 is the master (it owns the image, the optimizer state and
the regularizers, like the reference's parent process); every rank, rank 0 included, is a tile
worker.  Per evaluation:

    rank 0: cut the tiles of the (virtually rolled) image        style_transfer.py:623-637
    scatter: tile t -> rank t mod world   (the reference's round-robin, :284-288)
    all ranks: evaluate their tiles                               style_transfer.py:230-241
    gather: tile gradients and losses -> rank 0
    rank 0: stitch                                                style_transfer.py:639-643

Tiles never exchange data with each other, so there is no collective on the data path: the
scatter and the gather are batched point-to-point transfers (``batch_isend_irecv``), which on
RCCL run concurrently over the separate xGMI links from GPU 0 to each peer -- 12 bytes per tile
pixel each way.  The iteration's shift (``roll``) travels in the same batch as the tiles, as a
two-element message to each peer (the reference puts it into every SCGradRequest,
``style_transfer.py:158-161,634-637``).  Collectives run once, outside the step loop:
``broadcast_weights`` (the VGG filter bank, once per run) and ``broadcast_targets`` (style Grams
and content maps, once per scale); both hand back tensors on the compute device.  On RCCL a
target that already lives on rank 0's GPU (a ``DeviceArray`` or a CUDA tensor) is broadcast from
where it lies -- a 537 MB content map of a 4096 x 4096 scale never visits host memory; host arrays
(the small Grams) are uploaded once.  Only the gloo debug / CI wire stages through the host.  The arithmetic is injected through callables so that the protocol
can be exercised on CPU (gloo) in the unit tests.
"""

import numpy as np
import torch
import torch.distributed as dist


class DistributedTiles:
    """cut(rect, roll) -> tensor[3,th,tw] on rank 0; evaluate(jobs, roll) with
    jobs = [(tile tensor, (y0, x0)), ...] -> [(loss, grad tensor), ...] on every rank;
    put(rect, grad, roll) on rank 0.  All tensors live on ``device``.

    ``evaluate`` may also be a pair (begin, end): begin(jobs, roll) enqueues the work without
    waiting and end() -> [(loss, grad tensor), ...] waits for it.  Rank 0 then starts its own
    tiles while its sends to the other ranks are still in flight."""

    def __init__(self, cut, evaluate, put, device, group=None):
        self.cut, self.put = cut, put
        if callable(evaluate):
            self.evaluate, self.evaluate_begin, self.evaluate_end = evaluate, None, None
        else:
            self.evaluate_begin, self.evaluate_end = evaluate
            self.evaluate = lambda jobs, roll: (self.evaluate_begin(jobs, roll), self.evaluate_end())[1]
        self.device = torch.device(device)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._recv = {}
        # gloo moves host memory only: GPU tensors are staged through the CPU (debug / CI path;
        # on the GPU box the backend is nccl = RCCL and tensors travel device to device)
        self.host_staged = self.device.type == 'cuda' and dist.get_backend(group) == 'gloo'
        self.wire = torch.device('cpu') if self.host_staged else self.device

    def _sync(self):
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)

    def _buffer(self, key, shape, dtype=torch.float32):
        buf = self._recv.get(key)
        if buf is None or tuple(buf.shape) != tuple(shape):
            buf = self._recv[key] = torch.empty(shape, dtype=dtype, device=self.wire)
        return buf

    def _out(self, tensor):
        """A tensor as it goes on the wire."""
        return tensor.contiguous().to(self.wire) if self.host_staged else tensor.contiguous()

    def _in(self, tensor):
        """A received tensor as the callbacks want it (on the compute device)."""
        return tensor.to(self.device) if self.host_staged else tensor

    @staticmethod
    def _run(ops):
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def eval_sc_grad(self, rects, roll):
        """rects: the tile grid [(y0,y1,x0,x1)] (identical on all ranks); roll: (x, y) pixel
        shift, significant on rank 0.  Returns the summed loss on rank 0 (None elsewhere)."""
        owner = [t % self.world for t in range(len(rects))]
        mine = [t for t in range(len(rects)) if owner[t] == self.rank]
        shape = lambda t: (3, rects[t][1] - rects[t][0], rects[t][3] - rects[t][2])

        # ---- scatter
        tiles = {}
        ops = []
        if self.rank == 0:
            roll = (int(roll[0]), int(roll[1]))
            own = [t for t in range(len(rects)) if owner[t] == 0]
            # the shift goes to every peer that has work, in the same batch as its tiles
            header = torch.tensor(roll, dtype=torch.int64, device=self.wire)
            for r in sorted(set(owner) - {0}):
                ops.append(dist.P2POp(dist.isend, header, r, self.group))
            # peers' tiles first, so that they leave while this rank cuts and starts its own
            for t, rect in enumerate(rects):
                if owner[t] != 0:
                    ops.append(dist.P2POp(dist.isend, self._out(self.cut(rect, roll)), owner[t],
                                          self.group))
            self._sync()
            reqs = dist.batch_isend_irecv(ops) if ops else []
            for t in own:
                tiles[t] = self.cut(rects[t], roll)
            jobs = [(tiles[t], (rects[t][0], rects[t][2])) for t in mine]
            if self.evaluate_begin is not None and jobs:
                # no device-wide synchronisation here (it would wait for the sends): with a
                # two-phase evaluate, cut() must hand back finished tensors or run on the stream
                # evaluate_begin enqueues to
                self.evaluate_begin(jobs, roll)          # overlaps the sends
                for req in reqs:
                    req.wait()
                results = self.evaluate_end()
            else:
                for req in reqs:
                    req.wait()
                self._sync()
                results = self.evaluate(jobs, roll) if jobs else []
        else:
            header = self._buffer('roll', (2,), torch.int64)
            if mine:
                ops.append(dist.P2POp(dist.irecv, header, 0, self.group))
            for t in mine:
                tiles[t] = self._buffer(('tile', t), shape(t))
                ops.append(dist.P2POp(dist.irecv, tiles[t], 0, self.group))
            self._run(ops)
            if mine:
                hdr = header.cpu()
                roll = (int(hdr[0]), int(hdr[1]))
            tiles = {t: self._in(buf) for t, buf in tiles.items()}
            self._sync()
            # ---- evaluate the local tiles (possibly concurrently, that is the callee's business)
            results = self.evaluate([(tiles[t], (rects[t][0], rects[t][2])) for t in mine], roll) \
                if mine else []
        self._sync()

        # ---- gather
        ops = []
        if self.rank == 0:
            losses = [0.0] * len(rects)
            incoming = []
            for t in range(len(rects)):
                if owner[t] == 0:
                    continue
                gbuf = self._buffer(('grad', t), shape(t))
                ops.append(dist.P2POp(dist.irecv, gbuf, owner[t], self.group))
                incoming.append((t, gbuf))
            loss_bufs = {r: self._buffer(('loss', r), (sum(1 for o in owner if o == r),),
                                         torch.float64)
                         for r in range(1, self.world) if r in owner}
            for r, lbuf in loss_bufs.items():
                ops.append(dist.P2POp(dist.irecv, lbuf, r, self.group))
            self._run(ops)
            self._sync()
            for t, (loss, grad) in zip(mine, results):
                self.put(rects[t], grad, roll)
                losses[t] = loss
            for t, gbuf in incoming:
                self.put(rects[t], self._in(gbuf), roll)
            for r, lbuf in loss_bufs.items():
                for t, v in zip([t for t in range(len(rects)) if owner[t] == r], lbuf.tolist()):
                    losses[t] = v
            # added up in tile order, like the single-process farm does (the same bits)
            return float(sum(losses))
        if mine:
            for (loss, grad) in results:
                ops.append(dist.P2POp(dist.isend, self._out(grad), 0, self.group))
            losses = torch.tensor([loss for loss, _ in results], dtype=torch.float64,
                                  device=self.wire)
            ops.append(dist.P2POp(dist.isend, losses, 0, self.group))
            self._run(ops)
            self._sync()
        return None


def _wire_device(device, group):
    if torch.device(device).type == 'cuda' and dist.get_backend(group) == 'gloo':
        return torch.device('cpu')          # host-staged debug path, see DistributedTiles
    return torch.device(device)


def broadcast_weights(weights, device, group=None):
    """Rank 0's filter bank {conv layer: (w [Cout,Cin,k,k], b [Cout])} to every rank: ONE RCCL
    broadcast of the packed bank (80 MB for VGG-19), once per run (the reference makes every
    worker process read the .caffemodel itself, style_transfer.py:187-207).  Returns
    {layer: (w, b)} as views of one flat tensor on ``device``; keep the dict alive while engines
    copy from it."""
    rank = dist.get_rank(group)
    wire = _wire_device(device, group)
    meta = [None]
    if rank == 0:
        meta[0] = [(name, tuple(w.shape), tuple(b.shape)) for name, (w, b) in weights.items()]
    dist.broadcast_object_list(meta, 0, group=group)
    total = sum(int(np.prod(ws)) + int(np.prod(bs)) for _, ws, bs in meta[0])
    if rank == 0:
        flat = torch.from_numpy(np.concatenate(
            [np.concatenate([np.ravel(np.asarray(weights[name][0], np.float32)),
                             np.ravel(np.asarray(weights[name][1], np.float32))])
             for name, _, _ in meta[0]])).to(wire)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=wire)
    dist.broadcast(flat, 0, group=group)
    flat = flat.to(torch.device(device))
    out, pos = {}, 0
    for name, ws, bs in meta[0]:
        nw, nb = int(np.prod(ws)), int(np.prod(bs))
        out[name] = (flat[pos:pos + nw].view(ws), flat[pos + nw:pos + nw + nb].view(bs))
        pos += nw + nb
    return out


def broadcast_targets(contents, styles, device, group=None):
    """Broadcasts rank 0's targets (lists of {layer: array}) to every rank, once per scale.
    Sources may be numpy arrays, torch tensors or DeviceArrays (``__cuda_array_interface__``:
    broadcast in place on RCCL; ``.get()`` on the gloo debug wire); the result is lists of
    {layer: torch tensor on ``device``} -- the maps stay on the GPU.

    Ownership: on rank 0 a tensor made from a DeviceArray IS that array's memory (no copy: the
    broadcast reads it where it lies), so the caller keeps the DeviceArray alive for as long as it
    uses the returned tensor -- `DeviceArray.free()` / `StyleTransfer._drop_contents()` end both.
    The other ranks get allocations of their own."""
    rank = dist.get_rank(group)
    wire = _wire_device(device, group)
    meta = [None]
    if rank == 0:
        meta[0] = ([{k: tuple(v.shape) for k, v in c.items()} for c in contents],
                   [{k: tuple(v.shape) for k, v in s.items()} for s in styles])
    dist.broadcast_object_list(meta, 0, group=group)
    cshapes, sshapes = meta[0]
    out = []
    for shapes, src in ((cshapes, contents), (sshapes, styles)):
        items = []
        for i, shape_map in enumerate(shapes):
            d = {}
            for layer, shape in shape_map.items():
                if rank == 0:
                    v = src[i][layer]
                    if isinstance(v, torch.Tensor):
                        t = v.to(wire, torch.float32).contiguous()
                    elif hasattr(v, '__cuda_array_interface__') and wire.type == 'cuda':
                        # a DeviceArray on this rank's GPU: broadcast it from where it lies
                        if getattr(v, 'dtype', np.float32) != np.float32 or v.engine.device != (wire.index or 0):
                            raise ValueError('broadcast_targets: a float32 DeviceArray on this rank\'s GPU '
                                             'is expected for layer %s' % layer)
                        v.engine.sync()             # (written on the engine's stream, read on torch's)
                        t = torch.as_tensor(v, device=wire)
                    else:
                        v = v.get() if hasattr(v, 'get') else v
                        t = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(wire)
                else:
                    t = torch.empty(shape, dtype=torch.float32, device=wire)
                dist.broadcast(t, 0, group=group)
                d[layer] = t.to(torch.device(device))
            items.append(d)
        out.append(items)
    return out[0], out[1]
