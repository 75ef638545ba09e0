"""Convolution weights: seeded synthetic banks, ``.npz`` files and Caffe ``.caffemodel`` files.

The reference loads weights through ``caffe.Net(deploy, 1, weights=...)``
(``style_transfer.py:370``) from the files fetched by ``download_models.sh:5-6``.  No Caffe and
no ``.caffemodel`` exist offline, so this module reads the protobuf wire format directly
(``NetParameter``: new-style ``repeated LayerParameter layer = 100`` with ``name = 1``,
``type = 2``, ``blobs = 7``; legacy ``repeated V1LayerParameter layers = 2`` with ``name = 4``,
``blobs = 6``; ``BlobProto``: packed float ``data = 5``, ``shape = 7 { dim = 1 }`` or legacy
``num/channels/height/width = 1..4``) and can also write such a file, which is how the reader
is tested.  Benchmarks use ``synthetic_weights``: N(0, 2/(k*k*Cin)) filters and 0.01*N(0,1)
biases from a seeded ``RandomState``.
"""

import os
import struct

import numpy as np


def synthetic_weights(net, seed=0):
    """{conv layer name: (w [Cout,Cin,k,k], b [Cout])} float32, deterministic in ``seed``."""
    rng = np.random.RandomState(seed)
    channels = {net.input_blob(): net.layers[0].shape[1] if net.layers[0].shape else 3}
    params = {}
    for lay in net.layers[1:]:
        if lay.type == 'Convolution':
            cin, cout, k = channels[lay.bottom], lay.num_output, lay.kernel_size
            w = rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (k * k * cin))
            b = 0.01 * rng.standard_normal(cout)
            params[lay.name] = (w.astype(np.float32), b.astype(np.float32))
            channels[lay.top] = cout
        else:
            channels[lay.top] = channels[lay.bottom]
    return params


# ------------------------------------------------------------------ protobuf wire format ---
def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if not byte & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yields (field number, wire type, value) of one message; length-delimited values are
    memoryviews into ``buf``."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            value, pos = _varint(buf, pos)
        elif wire == 1:
            value, pos = buf[pos:pos + 8], pos + 8
        elif wire == 2:
            size, pos = _varint(buf, pos)
            value, pos = buf[pos:pos + size], pos + size
        elif wire == 5:
            value, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wire)
        yield field, wire, value


def _parse_blob(buf):
    dims, legacy, chunks = [], {}, []
    for field, wire, value in _fields(buf):
        if field == 5:                                   # data: packed or repeated float
            chunks.append(np.frombuffer(bytes(value), '<f4'))
        elif field == 7 and wire == 2:                   # shape { dim }
            for f2, w2, v2 in _fields(value):
                if f2 == 1 and w2 == 2:                  # packed int64 dims
                    p = 0
                    while p < len(v2):
                        d, p = _varint(v2, p)
                        dims.append(d)
                elif f2 == 1:
                    dims.append(v2)
        elif field in (1, 2, 3, 4) and wire == 0:        # legacy num / channels / height / width
            legacy[field] = value
    data = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
    if not dims and legacy:
        dims = [legacy.get(i, 1) for i in (1, 2, 3, 4)]
    return data, dims


def read_caffemodel(path):
    """{layer name: [blob arrays]} for every layer that carries blobs."""
    with open(path, 'rb') as f:
        buf = memoryview(f.read())
    out = {}
    for field, wire, value in _fields(buf):
        if wire != 2 or field not in (100, 2):
            continue
        name_field, blob_field = (1, 7) if field == 100 else (4, 6)
        name, blobs = None, []
        for f2, w2, v2 in _fields(value):
            if f2 == name_field and w2 == 2:
                name = bytes(v2).decode()
            elif f2 == blob_field and w2 == 2:
                data, dims = _parse_blob(v2)
                blobs.append(data.reshape(dims) if dims and int(np.prod(dims)) == data.size
                             else data)
        if name and blobs:
            out[name] = blobs
    return out


def _enc_varint(n):
    out = bytearray()
    while True:
        byte = n & 0x7F
        n >>= 7
        out.append(byte | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _enc_field(field, payload):
    return _enc_varint(field << 3 | 2) + _enc_varint(len(payload)) + payload


def write_caffemodel(path, params, legacy=False):
    """Writes {layer: (w, b)} as a NetParameter (new-style layers, or V1 ``layers`` if legacy)."""
    body = b''
    for name, blobs in params.items():
        layer = _enc_field(4 if legacy else 1, name.encode())
        if not legacy:
            layer += _enc_field(2, b'Convolution')
        for arr in blobs:
            arr = np.ascontiguousarray(arr, '<f4')
            if legacy:
                dims = ([1] * (4 - arr.ndim) + list(arr.shape))[-4:]
                blob = b''.join(_enc_varint(i << 3) + _enc_varint(d)
                                for i, d in zip((1, 2, 3, 4), dims))
            else:
                shape = _enc_field(1, b''.join(_enc_varint(d) for d in arr.shape))
                blob = _enc_field(7, shape)
            blob += _enc_field(5, arr.tobytes())
            layer += _enc_field(6 if legacy else 7, blob)
        body += _enc_field(2 if legacy else 100, layer)
    with open(path, 'wb') as f:
        f.write(body)


def load_weights(spec, net):
    """Resolves ``--weights``: 'synthetic' or 'synthetic:SEED', an ``.npz`` with
    ``<layer>_w`` / ``<layer>_b`` arrays, or a ``.caffemodel``."""
    spec = str(spec)
    if spec.startswith('synthetic'):
        _, _, seed = spec.partition(':')
        return synthetic_weights(net, int(seed) if seed else 0)
    if not os.path.isfile(spec):
        raise FileNotFoundError(
            "weights file '%s' not found (fetch it as the reference's download_models.sh does, "
            "or pass --weights synthetic for seeded random weights)" % spec)
    convs = [l for l in net.layers if l.type == 'Convolution']
    if spec.endswith('.npz'):
        z = np.load(spec)
        return {l.name: (np.float32(z[l.name + '_w']), np.float32(z[l.name + '_b']))
                for l in convs}
    blobs = read_caffemodel(spec)
    params = {}
    channels = {net.input_blob(): 3}
    for lay in net.layers[1:]:
        if lay.type != 'Convolution':
            channels[lay.top] = channels[lay.bottom]
            continue
        if lay.name not in blobs or len(blobs[lay.name]) < 2:
            raise KeyError("layer '%s' has no weights in %s" % (lay.name, spec))
        cin, k = channels[lay.bottom], lay.kernel_size
        w = np.float32(blobs[lay.name][0]).reshape(lay.num_output, cin, k, k)
        b = np.float32(blobs[lay.name][1]).reshape(lay.num_output)
        params[lay.name] = (w, b)
        channels[lay.top] = lay.num_output
    return params
