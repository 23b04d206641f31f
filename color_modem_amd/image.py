# -*- coding: utf-8 -*-
"""Frame adapter (API mirror of /root/reference/color_modem/image.py:11-84) plus batch entry points.

``ImageModem(modem).modulate(img, frame=0)`` / ``.demodulate(img, frame=0)`` take and return one
PIL image exactly like the reference.  ``demodulate_frames`` / ``modulate_frames`` are the batch
counterparts the GPU path sits behind: float32 planar frames in, float32 planar frames out,
equal to looping the reference's row schedule (image.py:47-55, 75-83) over
``frame = first_frame ..`` with a fresh modem per frame.
"""

import numpy

from color_modem_amd import engine as _engine


def _fused_u8(eng, direction):
    """Does this engine carry ImageModem's byte boundary inside its kernels for this direction?  An explicit answer (round 6; before, every
    NotImplementedError of the byte entry point - a pinned small-batch family, a capturing stream - silently switched to the float path with
    its own LSB rounding): engines say so through has_fused_u8(direction); the plan-based engines by their shape."""
    probe = getattr(eng, 'has_fused_u8', None)
    if probe is not None:
        return bool(probe(direction))
    return False


def _as_bytes(array):
    # same clamp + round-half-even as ref image.py:7-8
    return numpy.uint8(numpy.rint(255.0 * numpy.clip(array, 0.0, 1.0)))


class ImageModem(object):
    """batch_invariant (an addition to the reference's constructor): True pins the streaming kernels on whole rows for every batch size, so
    that a frame's result does not depend on the batch it arrives in - bit for bit, e.g. sharded against unsharded runs.  By default small
    batches take the row-parallel scan kernels or row segments (a picture in 18 instead of 200 microseconds), whose other operation
    order shows at float32 resolution (<= 2e-6 of full scale, SECAM <= 6e-6: DESIGN.md section 3.5)."""

    def __init__(self, modem, batch_invariant=False):
        self._modem = modem
        self._engine_obj = None
        self._batch_invariant = bool(batch_invariant)

    def _engine(self):
        if self._engine_obj is None:
            if not hasattr(self._modem, '_stack'):
                # not one of this package's modems: the reference drives any duck-typed object with modulate / demodulate (image.py:30, 49,
                # 54-55, 63, 77, 82-83) - so does this, row by row on the host (generic.RowLoopEngine), instead of an AttributeError
                from color_modem_amd import generic
                self._engine_obj = generic.RowLoopEngine(self._modem)
                return self._engine_obj
            self._engine_obj = _engine.make_engine(self._modem)
            if self._batch_invariant:
                self._engine_obj.set_small_batch('rows')
        return self._engine_obj

    @staticmethod
    def encode_composite_level(value):
        return 0.6 * value + 0.2

    @staticmethod
    def decode_composite_level(value):
        return (5.0 * value - 1.0) / 3.0

    # ---- batch API ------------------------------------------------------------------------------
    def demodulate_frames(self, composite, first_frame=0):
        """composite [F, H, W] float32 -> rgb [F, 3, H, W] float32 (numpy in -> numpy out, cuda tensor in -> cuda tensor out)."""
        return self._engine().demodulate_frames(composite, first_frame)

    def demodulate_frames_u8(self, composite8, first_frame=0):
        """composite uint8 [F, H, W] -> rgb uint8 [F, H, W, 3], the byte conversions of ImageModem fused into the kernel; for the stacks
        without a fused byte boundary (notches with a FilterFunction shift, avg= callables, widths that are not a multiple of 4) the same
        conversions run on the device around the float path (round 5: no host detour, identical bytes to the host-side formulas)."""
        eng = self._engine()
        if _fused_u8(eng, 'demod'):
            return eng.demodulate_frames_u8(composite8, first_frame)       # (a refusal of the native layer now surfaces instead of changing the path)
        return self._bytes_around_float(composite8, first_frame, demod=True)

    def modulate_frames(self, rgb, first_frame=0):
        """rgb [F, 3, H, W] float32 -> composite [F, H, W] float32."""
        return self._engine().modulate_frames(rgb, first_frame)

    def modulate_frames_u8(self, rgb8, first_frame=0):
        """rgb uint8 [F, H, W, 3] -> composite uint8 [F, H, W], the byte conversions of ImageModem fused into the kernel (widths that are not
        a multiple of 16, the noisy NIIR encoder: the same conversions on the device around the float path)."""
        eng = self._engine()
        if _fused_u8(eng, 'mod'):
            return eng.modulate_frames_u8(rgb8, first_frame)
        return self._bytes_around_float(rgb8, first_frame, demod=False)

    def _bytes_around_float(self, x8, first_frame, demod):
        """image.py:24-25, 47-55, 62, 75-83 as torch operations on the device, in float64 like the reference's numpy (so that every byte
        equals what the host-side formulas give), around the float entry points; frames in chunks."""
        import torch
        was_numpy = isinstance(x8, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(x8, dtype=numpy.uint8)) if was_numpy else x8
        if not torch.is_tensor(t) or t.dtype != torch.uint8 or t.dim() != (3 if demod else 4):
            raise ValueError('expected uint8 %s' % ('[n, H, W]' if demod else '[n, H, W, 3]'))
        if not t.is_cuda:
            t = t.cuda()
        n, h, w = int(t.shape[0]), int(t.shape[1]), int(t.shape[2])
        eng = self._engine()
        # picture and composite widths differ for some engines (MacEngine: 720 <-> 1080): the result is sized by the engine, not by the input
        w_out = int(getattr(eng, 'width', w)) if demod else int(getattr(eng, 'comp_width', w))
        out = torch.empty((n, h, w_out, 3) if demod else (n, h, w_out), dtype=torch.uint8, device=t.device)
        step = max(1, (1 << 28) // max(1, h * max(w, w_out) * 3 * 8))
        for f0 in range(0, n, step):
            part = t[f0:f0 + step]
            if demod:
                comp = self.decode_composite_level(part.double() / 255.0).float().contiguous()
                rgb = eng.demodulate_frames(comp, first_frame + f0)
                out[f0:f0 + step] = torch.round(255.0 * rgb.double().clamp(0.0, 1.0)).to(torch.uint8).permute(0, 2, 3, 1)
            else:
                rgb = (part.double() / 255.0).float().permute(0, 3, 1, 2).contiguous()
                comp = eng.modulate_frames(rgb, first_frame + f0)
                out[f0:f0 + step] = torch.round(255.0 * self.encode_composite_level(comp.double()).clamp(0.0, 1.0)).to(torch.uint8)
        return out.cpu().numpy() if was_numpy else out

    # ---- PIL API (one image = one frame) ---------------------------------------------------------
    def modulate(self, img, frame=0):
        from PIL import Image
        if img.mode != 'RGB':
            img = img.convert('RGB')
        rgb8 = numpy.frombuffer(img.tobytes(), dtype=numpy.uint8).reshape(img.height, img.width, 3)
        # the byte boundary fused into the kernel (widths that are multiples of 16), else the same conversions on the device around the float path
        comp8 = self.modulate_frames_u8(rgb8[None].copy(), frame)[0]
        return Image.frombytes('L', (comp8.shape[1], comp8.shape[0]), numpy.ascontiguousarray(comp8).tobytes())

    def demodulate(self, img, frame=0):
        from PIL import Image
        if img.mode != 'L':
            img = img.convert('L')
        comp8 = numpy.frombuffer(img.tobytes(), dtype=numpy.uint8).reshape(img.height, img.width).copy()
        # the byte boundary fused into the kernel (widths that are multiples of 4), else the same conversions on the device around the float path
        rgb8 = self.demodulate_frames_u8(comp8[None], frame)[0]
        return Image.frombytes('RGB', (rgb8.shape[1], rgb8.shape[0]), numpy.ascontiguousarray(rgb8).tobytes())
