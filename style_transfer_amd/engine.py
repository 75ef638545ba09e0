"""Python face of one libstx tile engine (one GPU).

Mirrors the server side of the reference's tile-worker protocol: the three requests a
``TileWorker`` answers (``style_transfer.py:215-259``) become three methods with the same
argument meaning --

    FeatureMapRequest(img, layers)                    -> features_tile(img, layers)
    SCGradRequest(img, roll, start, content_layers,   -> sc_grad_tile(img, start, roll, ...)
                  style_layers, dd_layers, layer_weights,
                  content_weight, style_weight, dd_weight)
    SetContentsAndStyles(contents, styles)            -> set_contents_and_styles(contents, styles)

-- and the arithmetic behind them (``CaffeModel.eval_features_tile`` / ``eval_sc_grad_tile``,
``style_transfer.py:421-427,556-612``) runs in the HIP kernels of libstx.  Arrays may be numpy
(host) or ``DeviceArray`` (resident on the engine's GPU).
"""

import ctypes

import numpy as np

from . import lib
from .netspec import NetSpec

_TYPE_CODES = {'Input': lib.LAYER_INPUT, 'Convolution': lib.LAYER_CONV, 'ReLU': lib.LAYER_RELU,
               'Pooling': lib.LAYER_POOL}


class DeviceArray:
    """A float32 (or uint8) array living on an engine's GPU."""

    def __init__(self, engine, shape, dtype=np.float32):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        ptr = ctypes.c_void_p()
        lib.call('stx_malloc', engine.handle, self.nbytes, ctypes.byref(ptr))
        self.ptr = ptr.value

    @classmethod
    def from_pointer(cls, engine, ptr, shape, dtype=np.float32, owner=None):
        """A view of device memory somebody else owns (e.g. a torch tensor: pass it as ``owner``
        to keep it alive).  ``free()`` does nothing."""
        arr = cls.__new__(cls)
        arr.engine, arr.shape, arr.dtype = engine, tuple(int(v) for v in shape), np.dtype(dtype)
        arr.nbytes = int(np.prod(arr.shape, dtype=np.int64)) * arr.dtype.itemsize
        arr.ptr, arr.owner = int(ptr), owner
        arr.free = lambda: None
        return arr

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def __cuda_array_interface__(self):
        """Lets torch (``torch.as_tensor(arr, device=...)``) view the array without a copy, e.g.
        as the source of an RCCL broadcast.  The caller orders the streams (``engine.sync()``)."""
        return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (int(self.ptr), False),
                'version': 2, 'strides': None}

    def set(self, host):
        host = np.ascontiguousarray(host, self.dtype)
        assert host.shape == self.shape, (host.shape, self.shape)
        lib.call('stx_memcpy_async', self.engine.handle, self.ptr, lib.DEVICE, host.ctypes.data,
                 lib.HOST, self.nbytes)
        self.engine.sync()
        return self

    def copy_from(self, other):
        assert other.nbytes == self.nbytes
        lib.call('stx_memcpy_async', self.engine.handle, self.ptr, lib.DEVICE, other.ptr,
                 lib.DEVICE, self.nbytes)
        return self

    def zero(self):
        lib.call('stx_memset_async', self.engine.handle, self.ptr, 0, self.nbytes)
        return self

    def get(self):
        out = np.empty(self.shape, self.dtype)
        lib.call('stx_memcpy_async', self.engine.handle, out.ctypes.data, lib.HOST, self.ptr,
                 lib.DEVICE, self.nbytes)
        self.engine.sync()
        return out

    def free(self):
        if self.ptr and self.engine.handle:
            lib.call('stx_free', self.engine.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # pylint: disable=broad-except
            pass


def _as_arg(arr):
    """(pointer, mem tag, keep-alive object) for a numpy array, a DeviceArray or a float32 torch
    tensor (on a GPU: used in place; on the CPU: its memory is the host buffer)."""
    if isinstance(arr, DeviceArray):
        return arr.ptr, lib.DEVICE, arr
    if hasattr(arr, 'data_ptr') and hasattr(arr, 'is_cuda'):        # torch.Tensor, no import
        t = arr.contiguous()
        if str(t.dtype) != 'torch.float32':
            t = t.float()
        return t.data_ptr(), (lib.DEVICE if t.is_cuda else lib.HOST), t
    host = np.ascontiguousarray(arr, np.float32)
    return host.ctypes.data, lib.HOST, host


class PendingTile:
    """Result of an asynchronous sc_grad_tile: valid after ``engine.sync()``."""

    def __init__(self, grad, keep):
        self._loss = ctypes.c_double(float('nan'))
        self.grad = grad
        self._keep = keep

    @property
    def loss(self):
        return self._loss.value


class TileEngine:
    """One GPU's worker: the network, its weights and the current targets."""

    def __init__(self, net, device=0, weights=None, share=None):
        """``share``: an engine on the same GPU whose weights, packed filter banks and targets
        this one uses (stx_engine_create_shared); ``weights`` is then ignored."""
        assert isinstance(net, NetSpec)
        self.net = net
        self.device = device
        self.handle = None
        self._names = {}
        self.primary = share.primary if share is not None else self
        handle = ctypes.c_void_p()
        if share is not None:
            assert share.device == device and share.net is net
            lib.call('stx_engine_create_shared', share.handle, ctypes.byref(handle))
        else:
            descs = (lib.LayerDesc * len(net.layers))()
            for d, lay in zip(descs, net.layers):
                d.name = self._cstr(lay.name)
                d.type = _TYPE_CODES[lay.type]
                d.bottom = self._cstr(lay.bottom) if lay.bottom else None
                d.top = self._cstr(lay.top)
                d.num_output = lay.num_output if lay.type != 'Input' else \
                    (lay.shape[1] if lay.shape else 3)
                d.kernel_size, d.pad, d.stride = lay.kernel_size, lay.pad, lay.stride
                d.pool_mode = lib.POOL_AVE if lay.pool == 'AVE' else lib.POOL_MAX
            lib.call('stx_engine_create', device, descs, len(net.layers), ctypes.byref(handle))
        self.handle = handle
        self._results = []
        self._info = {b: net.layer_info(b) for b in net.blob_names()}
        self.n_styles = 0
        if weights and share is None:
            for name, (w, b) in weights.items():
                self.set_weights(name, w, b)

    def _cstr(self, s):
        """Encoded layer name, kept alive for as long as the engine (one entry per distinct
        name: tap tables are rebuilt on every call)."""
        b = self._names.get(s)
        if b is None:
            b = self._names[s] = s.encode()
        return b

    def close(self):
        if self.handle:
            lib.load().stx_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass

    # ------------------------------------------------------------------------------ basics
    def sync(self):
        lib.call('stx_sync', self.handle)
        self._results = []
        self._fence_marks = {}
        self._dropped = 0

    def keep_until_sync(self, result):
        """Holds a reference to an object the library will write into at the next sync (a
        pending loss): it must not be collected before, even if its owner lets go of it.  The
        library publishes pending values whenever it drains its scalar arena, also inside calls
        that never come back through ``sync()``.  Entries are dropped when their values have been
        published: at ``sync()`` and at ``wait_fence()`` (everything queued before that fence).  The
        count below is a backstop for callers that never do either -- the library drains its arena
        every few hundred evaluations, so an entry 2048 evaluations old has long been written.  (Each
        entry pins a tile-sized host buffer: the cap stays small.)"""
        self._results.append(result)
        if len(self._results) > 4096:
            del self._results[:2048]
            self._dropped = getattr(self, '_dropped', 0) + 2048
        return result

    def fence(self):
        """Closes the pending values queued so far behind an event and returns a ticket for
        ``wait_fence`` (stx_fence): the host can queue the next iteration first and collect this
        one's losses afterwards, without waiting for the newer work."""
        ticket = ctypes.c_ulonglong(0)
        lib.call('stx_fence', self.handle, ctypes.byref(ticket))
        if not hasattr(self, '_fence_marks'):
            self._fence_marks, self._dropped = {}, 0
        self._fence_marks[ticket.value] = self._dropped + len(self._results)
        return ticket.value

    def wait_fence(self, ticket):
        """Waits for the fence's event only and publishes the values it closed; the objects
        those values were written into are released from ``keep_until_sync``."""
        lib.call('stx_fence_wait', self.handle, int(ticket))
        mark = getattr(self, '_fence_marks', {}).pop(int(ticket), None)
        if mark is not None:
            n = mark - self._dropped
            if n > 0:
                del self._results[:n]
                self._dropped += n

    def wait_for(self, other):
        """Orders this engine's stream behind what ``other`` has queued so far (no host wait)."""
        if other is not self:
            lib.call('stx_engine_wait', self.handle, other.handle)

    def query(self, what):
        """One of the lib.Q_* counters."""
        v = ctypes.c_double(0)
        lib.call('stx_engine_query', self.handle, int(what), ctypes.byref(v))
        return v.value

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def to_device(self, host, dtype=np.float32):
        host = np.ascontiguousarray(host, dtype)
        return DeviceArray(self, host.shape, dtype).set(host)

    def layer_info(self, layer):
        """(scale, channels) -- CaffeModel.layer_info, style_transfer.py:415-419."""
        return self._info[layer]

    def feature_shape(self, layer, th, tw):
        if layer not in self._info:
            raise lib.StxError('feature_shape', -1, "unknown layer '%s'" % layer)
        scale, ch = self._info[layer]
        h, w = th, tw
        s = 1
        while s < scale:          # one ceil-mode halving per pooling stage
            h, w, s = (h + 1) // 2, (w + 1) // 2, s * 2
        return ch, h, w

    def set_weights(self, conv_layer, w, b):
        """w [Cout,Cin,k,k], b [Cout]: numpy, DeviceArray or torch tensors (device-resident
        sources are copied on the engine's stream, e.g. straight out of an RCCL broadcast)."""
        wp, wmem, wkeep = _as_arg(w)
        bp, bmem, bkeep = _as_arg(b)
        if wmem != bmem:
            raise ValueError('weights and bias must live on the same side')
        lib.call('stx_set_conv_weights', self.handle, conv_layer.encode(), wp, bp, wmem)
        if wmem == lib.DEVICE:
            self.sync()          # the sources may be released once this returns
        del wkeep, bkeep

    # ------------------------------------------------------------- SetContentsAndStyles
    def set_contents_and_styles(self, contents, styles):
        """contents: list of {layer: [C,h,w] full-image feature map}; styles: list of
        {layer: [C,C] lower-triangular Gram} (style_transfer.py:243-254)."""
        keep = []
        n_c = sum(len(c) for c in contents)
        n_s = sum(len(s) for s in styles)
        ctargets = (lib.ContentTarget * max(1, n_c))()
        starget = (lib.StyleTarget * max(1, n_s))()
        i = 0
        for ci, content in enumerate(contents):
            for layer, feat in content.items():
                ptr, mem, obj = _as_arg(feat)
                keep.append(obj)
                t = ctargets[i]
                t.content_index, t.layer = ci, self._cstr(layer)
                t.channels, t.height, t.width = obj.shape
                t.features, t.mem = ptr, mem
                i += 1
        i = 0
        for si, style in enumerate(styles):
            for layer, gram in style.items():
                ptr, mem, obj = _as_arg(gram)
                keep.append(obj)
                t = starget[i]
                t.style_index, t.layer, t.channels = si, self._cstr(layer), obj.shape[0]
                t.gram, t.mem = ptr, mem
                i += 1
        lib.call('stx_set_contents_and_styles', self.handle, ctargets, n_c, starget, n_s)
        self.n_styles = len(styles)

    # --------------------------------------------------------------------- FeatureMapRequest
    def features_tile(self, img, layers):
        """Post-ReLU feature maps of one tile: {layer: [C, ceil(th/s), ceil(tw/s)] ndarray}."""
        ptr, mem, keep = _as_arg(img)
        th, tw = keep.shape[-2:]
        outs = [np.empty(self.feature_shape(l, th, tw), np.float32) for l in layers]
        names = (ctypes.c_char_p * len(layers))(*[l.encode() for l in layers])
        ptrs = (ctypes.c_void_p * len(layers))(*[o.ctypes.data for o in outs])
        lib.call('stx_features_tile', self.handle, ptr, mem, th, tw, names, len(layers), ptrs,
                 lib.HOST)
        self.sync()
        return dict(zip(layers, outs))

    def features_tile_device(self, img, layers, out=None):
        """Like features_tile, but the maps stay on the GPU: {layer: DeviceArray}.  ``out`` may
        supply the destination arrays (reused across tiles).  Asynchronous."""
        ptr, mem, keep = _as_arg(img)
        th, tw = keep.shape[-2:]
        if out is None:
            out = {}
        for l in layers:
            shape = self.feature_shape(l, th, tw)
            if l not in out or out[l].shape != shape:
                out[l] = self.empty(shape)
        names = (ctypes.c_char_p * len(layers))(*[l.encode() for l in layers])
        ptrs = (ctypes.c_void_p * len(layers))(*[out[l].ptr for l in layers])
        lib.call('stx_features_tile', self.handle, ptr, mem, th, tw, names, len(layers), ptrs,
                 lib.DEVICE)
        if mem == lib.HOST:
            self.sync()
        return out

    def map_place(self, dst, y0, x0, src):
        """dst[:, y0:y0+h, x0:x0+w] = src on the device (feature-map stitch)."""
        c, h, w = src.shape
        lib.call('stx_map_place', self.handle, dst.ptr, c, dst.shape[1], dst.shape[2], int(y0),
                 int(x0), src.ptr, h, w)

    def map_roll_add(self, acc, src, roll_xy, alpha, init_divisor=0.0):
        """acc = roll2(src, roll_xy) / init_divisor, or acc += alpha * roll2(src, roll_xy)."""
        c, h, w = src.shape
        roll = (ctypes.c_int * 2)(int(roll_xy[0]), int(roll_xy[1]))
        lib.call('stx_map_roll_add', self.handle, acc.ptr, src.ptr, c, h, w, roll, float(alpha),
                 float(init_divisor))

    # --------------------------------------------------------------------------- SCGradRequest
    def _taps(self, content_layers, style_layers, layer_weights, content_weight, style_weight,
              dd_layers=(), dd_weight=None):
        names = list(dict.fromkeys(list(content_layers) + list(style_layers) + list(dd_layers)))
        taps = (lib.Tap * len(names))()
        for t, name in zip(taps, names):
            t.layer = self._cstr(name)
            t.layer_weight = float(layer_weights.get(name, 1.0)) if layer_weights else 1.0
            t.is_content = int(name in content_layers)
            t.content_weight = float(content_weight.get(name, 0.0)) if t.is_content else 0.0
            t.is_style = int(name in style_layers)
            t.style_weight = float(style_weight.get(name, 0.0)) if t.is_style else 0.0
            t.is_dd = int(name in dd_layers)
            t.dd_weight = float(dd_weight.get(name, 0.0)) if t.is_dd and dd_weight else 0.0
        return taps, len(names)

    def io_buffers(self, th, tw):
        """(tile, gradient) DeviceArray views of the engine's own input blob and of the blob its
        gradient is left in: a tile written into the first and evaluated with grad_out = the
        second needs no copies (stx_tile_buffers).  Query before every use."""
        tin, gout = ctypes.c_void_p(), ctypes.c_void_p()
        lib.call('stx_tile_buffers', self.handle, int(th), int(tw), ctypes.byref(tin), ctypes.byref(gout))
        return (DeviceArray.from_pointer(self, tin.value, (3, th, tw)),
                DeviceArray.from_pointer(self, gout.value, (3, th, tw)))

    def sc_grad_tile_async(self, img, start, roll, content_layers, style_layers, layer_weights,
                           content_weight, style_weight, grad_out=None, dd_layers=(),
                           dd_weight=None):
        """Enqueues one tile evaluation; returns a PendingTile (read it after ``sync()``)."""
        ptr, mem, keep = _as_arg(img)
        th, tw = keep.shape[-2:]
        if grad_out is None:
            grad_out = np.empty((3, th, tw), np.float32)
        gptr, gmem, gkeep = (grad_out.ptr, lib.DEVICE, grad_out) \
            if isinstance(grad_out, DeviceArray) else (grad_out.ctypes.data, lib.HOST, grad_out)
        taps, n_taps = self._taps(content_layers, style_layers, layer_weights, content_weight,
                                  style_weight, dd_layers, dd_weight)
        roll_c = (ctypes.c_int * 2)(int(roll[0]), int(roll[1])) if roll is not None \
            else (ctypes.c_int * 2)(0, 0)
        start_c = (ctypes.c_int * 2)(int(start[0]), int(start[1]))
        pending = self.keep_until_sync(PendingTile(gkeep, (keep, taps)))
        lib.call('stx_sc_grad_tile', self.handle, ptr, mem, th, tw, roll_c, start_c, taps, n_taps,
                 ctypes.byref(pending._loss), gptr, gmem, 0)
        return pending

    def sc_grad_tile(self, img, start, roll, content_layers, style_layers, layer_weights,
                     content_weight, style_weight, dd_layers=(), dd_weight=None):
        """(loss, grad[3,th,tw]) of one tile -- CaffeModel.eval_sc_grad_tile with the worker's
        content roll (style_transfer.py:230-241,556-612)."""
        pending = self.sc_grad_tile_async(img, start, roll, content_layers, style_layers,
                                          layer_weights, content_weight, style_weight,
                                          dd_layers=dd_layers, dd_weight=dd_weight)
        self.sync()
        return pending.loss, pending.grad

    def gram_matrix(self, feat):
        """Lower-triangular Gram of a [C,h,w] feature map (num_utils.py:143-147)."""
        ptr, mem, keep = _as_arg(feat)
        c = keep.shape[0]
        hw = int(np.prod(keep.shape[1:]))
        out = np.empty((c, c), np.float32)
        lib.call('stx_gram_matrix', self.handle, ptr, mem, c, hw, out.ctypes.data, lib.HOST)
        return out

    def profile(self, on=True):
        """Turns per-kernel-group event timing on or off (clears the record)."""
        lib.call('stx_profile_enable', self.handle, int(on))

    def profile_read(self, clock=False):
        """[(label, milliseconds, algorithmic flops)] recorded since the last read; with
        ``clock=True`` a fourth field: the shader clock in MHz inside the group's convolution kernel
        (0 unless clock_marks is on and the group is a 2-D Winograd launch)."""
        buf = ctypes.create_string_buffer(1 << 20)
        lib.call('stx_profile_read', self.handle, buf, len(buf), None)
        rows = []
        for line in buf.value.decode().splitlines():
            label, ms, flops, mhz = line.split('\t')
            rows.append((label, float(ms), float(flops), float(mhz)) if clock else
                        (label, float(ms), float(flops)))
        return rows

    def last_tile_flops(self):
        """(algorithmic, issued) matrix-core FLOP of the convolutions of the last tile call."""
        a, b = ctypes.c_double(0), ctypes.c_double(0)
        lib.call('stx_last_tile_flops', self.handle, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def clock_marks(self, on):
        """While on, one workgroup of every 2-D Winograd convolution launch of this engine times its
        chunk loop with the core-cycle counter and the constant 100 MHz counter (stx_clock_marks):
        the shader clock inside the kernel that does most of the work."""
        lib.call('stx_clock_marks', self.handle, 1 if on else 0)

    def clock_marks_read(self, max_values=16384):
        """MHz of the marks recorded since the last read, launch order (0: a loop too short to
        tell); synchronises this engine's stream and clears the record."""
        mhz = (ctypes.c_double * max_values)()
        n = ctypes.c_int(0)
        lib.call('stx_clock_marks_read', self.handle, mhz, max_values, ctypes.byref(n))
        return [mhz[i] for i in range(n.value)]

    def last_tile_ms(self):
        ms = ctypes.c_float(0)
        lib.call('stx_last_tile_ms', self.handle, ctypes.byref(ms))
        return ms.value
