import sys, numpy
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import am_stacks, stacks
from color_modem_amd import image, line, testing
from oracle import cm_oracle_am as oa
for stack,size,std,first in [('niir', (720, 64), 'GERBER_625', 2), ('niir_hue', (720, 33), 'GERBER_625', 1), ('niir', (768, 9), 'NTSC_525', 4798), ('niir', (718, 12), 'GERBER_625', 5), ('niir', (1024, 40), 'GERBER_625', 0), ('niir', (720, 576), 'GERBER_625', 3)]:
    lc = line.LineConfig(size, getattr(line.LineStandard, std))
    modem = am_stacks.STACKS[stack](lc)
    rgb = testing.synthetic_rgb(2, size[1], size[0], seed=55 + size[1])
    im = image.ImageModem(modem)
    comp_ref = oa.modulate_frames(modem, rgb.astype(numpy.float64), first)
    comp32 = comp_ref.astype(numpy.float32)
    back = im.demodulate_frames(comp32, first_frame=first)
    back_ref = oa.demodulate_frames(modem, comp32.astype(numpy.float64), first)
    for i in range(2):
        err = numpy.abs(back[i] - back_ref[i]) / numpy.abs(back_ref[i]).max()
        ix = numpy.unravel_index(err.argmax(), err.shape)
        cols = err.max(axis=(0,1))
        print(stack, size, 'frame', i, 'max %.2e at %s  q99.99 %.2e  cols>1e-5: %s' % (err.max(), ix, numpy.quantile(err, 0.9999), numpy.nonzero(cols > 1e-5)[0][:20]), flush=True)
