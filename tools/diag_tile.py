"""Order / repeatability diagnostic for the tile path (GPU box)."""
import numpy as np, sys
sys.path.insert(0,'.')
from tests.gpu_helpers import gpu_engine
from tests.helpers import *
from tests.test_gpu_tile_path import _targets
z=np.load('tests/golden/reference_vectors.npz'); golden={k:z[k] for k in z.files}
T={}
for tag,model in [('vgg16avg','vgg16_avgpool'),('vgg19','vgg19')]:
    T[tag]=_targets(golden, tag, model)
for tag,model in [('vgg16avg','vgg16_avgpool'),('vgg19','vgg19'),('vgg19','vgg19'),('vgg16avg','vgg16_avgpool')]:
    g, om, (cl,cw,sl,sw) = T[tag]
    eng = gpu_engine(model)
    eng.set_contents_and_styles(om.contents, om.styles)
    lw={'conv3_1':1.5}
    tile=np.ascontiguousarray(g['img_rolled'][:,8:48,16:72])
    for rep in range(2):
        loss,grad=eng.sc_grad_tile(tile,(8,16),(0,0),cl,sl,lw,cw,sw)
        ref=g['single.grad']
        print(tag,rep,'l2rel',np.linalg.norm(grad-ref)/np.linalg.norm(ref))
    # drop terms one at a time
    for drop in sl+cl:
        sl2=[s for s in sl if s!=drop]; cl2=[c for c in cl if c!=drop]
        l1,g1=eng.sc_grad_tile(tile,(8,16),(0,0),cl2,sl2,lw,cw,sw)
        l2,g2=om.sc_grad_tile(tile,(8,16),cl2,sl2,lw,cw,sw)
        print('   without',drop,'l2rel',np.linalg.norm(g1-g2)/np.linalg.norm(g2), l1, l2)
