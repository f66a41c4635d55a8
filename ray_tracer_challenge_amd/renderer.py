"""Device-resident rendering: scene uploaded once, output left in HBM.

`Renderer` wraps the persistent-context half of the C ABI (rtc_ctx_*).
PyTorch appears here only as plumbing -- it owns the output tensor and the
stream, and (in dist.py) carries the RCCL gather.  The kernel launch itself goes
through librtc_amd.so with raw device pointers.
"""
import ctypes as C

import torch

from . import _lib as L


class Renderer:
    def __init__(self, world, camera, device=None):
        if not torch.cuda.is_available():
            raise L.RtcError(L.RTC_ERR_NO_DEVICE, "no GPU visible to torch; the render path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.width, self.height = camera.width, camera.height
        self._ctx = C.c_void_p()
        self._lib = L.lib()  # the library that owns this context (tests load a second one beside it: _lib.use_library)
        L.check(self._lib.rtc_ctx_create(self.device.index, C.byref(self._ctx)))
        self._keep = None
        self.set_scene(world, camera)

    def set_scene(self, world, camera):
        cs = world._c()
        self._keep = (cs, camera)
        L.check(self._lib.rtc_ctx_set_scene(self._ctx, C.byref(cs.scene), C.byref(camera._cam)))
        self.width, self.height = camera.width, camera.height

    def set_camera(self, camera):
        """Another camera on the world that is resident (an animation's usual frame): the world is not flattened again on the
        Python side, and the library, finding the records unchanged, uploads none of them."""
        cs = self._keep[0]
        self._keep = (cs, camera)
        L.check(self._lib.rtc_ctx_set_scene(self._ctx, C.byref(cs.scene), C.byref(camera._cam)))
        self.width, self.height = camera.width, camera.height

    def close(self):
        if self._ctx:
            self._lib.rtc_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def partition(band_rows=64, n_parts=1, part=0):
        return L.rtc_partition(band_rows, n_parts, part)

    def rows(self, part=None):
        return int(self._lib.rtc_partition_rows(self.height, C.byref(part) if part is not None else None))

    def alloc(self, part=None):
        return torch.empty((self.rows(part), self.width, 3), dtype=torch.float32, device=self.device)

    def render(self, depth, out=None, part=None, stream=None):
        """Launches the render kernel on `stream` (default: torch's current stream); asynchronous."""
        if out is None:
            out = self.alloc(part)
        # (checked, not asserted: the raw pointer goes to a kernel that writes rows x width x 3 floats through it)
        if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == self.rows(part) * self.width * 3):
            raise ValueError("out must be a contiguous float32 CUDA tensor of %d x %d x 3" % (self.rows(part), self.width))
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        L.check(self._lib.rtc_ctx_render(self._ctx, int(depth), C.byref(part) if part is not None else None,
                                       C.c_void_p(out.data_ptr()), C.c_void_p(s.cuda_stream)))
        return out

    @property
    def kernel_name(self):
        return self._lib.rtc_ctx_kernel_name(self._ctx).decode()

    @property
    def kernel_id(self):
        """Names the code object that renders the current scene (rtc_ctx_kernel_id)."""
        return self._lib.rtc_ctx_kernel_id(self._ctx).decode()

    @property
    def jit_status(self):
        """"" when the scene's kernel is what the specialisation policy asked for, else the reason (rtc_ctx_jit_status)."""
        return self._lib.rtc_ctx_jit_status(self._ctx).decode(errors="replace")

    def stats(self):
        """Synchronises with the last render and returns its counters."""
        st = L.rtc_stats()
        L.check(self._lib.rtc_ctx_stats(self._ctx, C.byref(st)))
        return {"rays": int(st.rays), "shaded_hits": int(st.shaded_hits), "pixels": int(st.pixels),
                "kernel_ms": float(st.kernel_ms), "launches": int(st.launches), "rows": int(st.rows),
                "culled_shadow_rays": int(st.culled_shadow_rays), "flags": int(st.flags)}

    def to_ppm(self, rgb, stream=None):
        """Canvas::to_ppm (canvas.rs:58-96) formatted on the device from an (h, w, 3) f32 tensor -> bytes."""
        if not (rgb.is_cuda and rgb.dtype == torch.float32 and rgb.is_contiguous() and rgb.dim() == 3):
            raise ValueError("rgb must be a contiguous float32 CUDA tensor (rows, width, 3)")
        h, w = int(rgb.shape[0]), int(rgb.shape[1])
        cap = int(self._lib.rtc_ppm_max_bytes(w, h))
        text = torch.empty(cap, dtype=torch.uint8, device=rgb.device)
        n = C.c_uint64()
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        L.check(self._lib.rtc_ctx_to_ppm(self._ctx, C.c_void_p(rgb.data_ptr()), w, h, C.c_void_p(text.data_ptr()), cap,
                                       C.byref(n), C.c_void_p(s.cuda_stream)))
        return text[: n.value].cpu().numpy().tobytes()

    def quantize(self, rgb, stream=None, out=None):
        """canvas.rs:39-43 scale_color on the device: f32 tensor -> u8 tensor of the same shape."""
        if out is None:
            out = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
        if not (rgb.is_cuda and rgb.dtype == torch.float32 and rgb.is_contiguous() and out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and
                out.numel() == rgb.numel()):
            raise ValueError("rgb: a contiguous float32 CUDA tensor; out: a contiguous uint8 CUDA tensor with one byte per colour value of rgb")
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        L.check(self._lib.rtc_ctx_quantize(self._ctx, C.c_void_p(rgb.data_ptr()), rgb.numel(),
                                         C.c_void_p(out.data_ptr()), C.c_void_p(s.cuda_stream)))
        return out
