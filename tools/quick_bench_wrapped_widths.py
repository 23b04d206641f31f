import sys, time, torch, numpy
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import stacks
from color_modem_amd import image, testing
NAMES = sys.argv[1:] or ['simple3d_pald']
for name, size in [(n, sz) for n in NAMES for sz in ((768,576), (1280,576), (1920,576))]:
    F = 1000 if size[0] < 1000 else 400
    m = stacks.make(name, size)
    eng = image.ImageModem(m)._engine()
    comp = torch.rand((F, size[1], size[0]), device='cuda')*0.6+0.2
    out = torch.empty((F,3,size[1],size[0]), device='cuda')
    for mode in ('fused','composition'):
        if mode=='composition':
            eng.set_small_batch('rows')
        eng.demodulate_frames(comp, 0, out=out); torch.cuda.synchronize()
        ts=[]
        for _ in range(5):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(); eng.demodulate_frames(comp,0,out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        t=sorted(ts)[2]
        print('%s %dx%d x %d frames %-12s %.3f ms  %.1f Gpx/s   [%s]' % (name, size[0], size[1], F, mode, t, F*size[0]*size[1]/t/1e6, 'fused plan' if eng.fused is not None else 'no fused plan'), flush=True)
