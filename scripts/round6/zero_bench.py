import ctypes as C, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from warpx_amd import load_product, _capi
from warpx_amd.containers import FieldArray
lib = load_product()
fs = [FieldArray((256,256,256), s, (5,5,5), "cuda") for s in ((0,1,1),(1,0,1),(1,1,0))]
views = (_capi.FieldView * 3)(*[f.view for f in fs])
for name, fn in (("zero_multi", lambda: lib.field_set_zero_multi(views, 3, None)),
                 ("3 x memset", lambda: [lib.field_set_zero(C.byref(f.view), None) for f in fs])):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); e1.synchronize()
    print(name, "%.1f us per call" % (e0.elapsed_time(e1) / 50 * 1e3))
