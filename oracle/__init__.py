"""CPU oracle for the tiled style-transfer hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  Nothing under ``style_transfer_amd/``
imports it, and the product path fails loudly when the HIP library is missing
instead of falling back to anything here.

It is a numpy restatement of the reference's per-tile path
(crowsonkb/style_transfer: ``style_transfer.py:421-427,556-645``,
``num_utils.py:20-162``, ``optimizers.py:11-138``) plus the arithmetic of the
BVLC/caffe layers the reference delegates to (Convolution / in-place ReLU /
ceil-mode Pooling; un-vendored and unpinned: ``docker/Dockerfile:29``).

Parity pin: the reference ships no tests or golden vectors.  The restatement is
pinned against fixtures under ``tests/golden/`` that were generated in the build
container by executing the reference's own Python (``CaffeModel.eval_sc_grad_tile``
etc., unmodified, imported from /root/reference) over a pycaffe-shaped shim
(``oracle/caffe_net.py``); see ``tests/golden/make_golden.py``.  The Caffe layer
arithmetic itself has no upstream pin ("parity unpinned" for BVLC/caffe): it is
restated from Caffe's documented semantics and cross-checked against torch-CPU
``conv2d`` / ``max_pool2d(ceil_mode=True)`` autograd in ``tests/``.
"""
