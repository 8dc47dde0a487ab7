import sys, json
sys.path.insert(0,'.')
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
SERVER = len(sys.argv) > 1 and sys.argv[1] == 'server'     # round-1..5 stand-in for C3: widened LCNet det + SVTR-neck rec (V = 18710), 1280^2 pages
C3 = len(sys.argv) > 1 and sys.argv[1] == 'c3'             # BASELINE C3 on the graphs it names: PP-HGNetV2 / LK-PAN detector + SVTRv2 recognizer
if C3:
    det,_=models.build_det('server_hgnet', seed=0); rec,_=models.build_rec('svtrv2', vocab=6625, seed=1)
    chars=api.read_dict(models.synth_dict(6623)); S=1280; NP=int(sys.argv[2]) if len(sys.argv) > 2 else 16
elif SERVER:
    det,_=models.build_det('server', seed=0); rec,_=models.build_rec('server', vocab=18710, seed=1)
    chars=api.read_dict(models.synth_dict(18708)); S=1280; NP=16
elif len(sys.argv) > 1 and sys.argv[1] == 'full':         # BASELINE C2 on the graphs bench.py times by default since round 6 (0.447 M / 1.103 M parameters)
    det,_=models.build_det('tiny_full', seed=0); rec,_=models.build_rec('tiny_full', vocab=6906, seed=1)
    chars=api.read_dict(models.synth_dict()); S=960; NP=32
else:
    det,_=models.build_det(); rec,_=models.build_rec()
    chars=api.read_dict(models.synth_dict()); S=960; NP=32
P=[pages.make_page(i,(S,S),40) for i in range(NP)]
bufs=[api.DeviceBuffer(p) for p in P]
ocr=api.OAROCRBuilder(det,rec,chars).text_detection_config(api.TextDetectionConfig(0.3,0.6,1.5,limit_side_len=S)).image_batch_size(32).region_batch_size(256).build()
ptrs=[int(b.ptr.value) for b in bufs]
ocr.predict_device(ptrs,[S]*NP,[S]*NP,raw=True)
api.prof_enable(True); api.prof_reset()
import time
t=time.perf_counter(); r=ocr.predict_device(ptrs,[S]*NP,[S]*NP,raw=True); dt=time.perf_counter()-t
snap=api.prof_snapshot()
print('step ms',dt*1e3, r)
tot=sum(e['total_ms'] for e in snap)
print('total kernel ms',tot, 'classes',len(snap))
for e in snap:
    ms=e['total_ms']; 
    print(f"{e['name']:60s} n={e['launches']:4d} ms={ms:8.3f} us/launch={ms*1e3/max(e['launches'],1):8.1f} GB/s={e['alg_bytes']/ms/1e6 if ms else 0:8.1f} TF={e['alg_flops']/ms/1e9 if ms else 0:6.2f}")
