"""MI355X-native area-average image interpolation: Python plumbing over the C ABI in include/aai.h."""
from . import _lib
from ._lib import (DTYPE_F32, DTYPE_U8, DTYPE_U16, MODE_AREA, MODE_FAST, MODE_BILINEAR, MODE_BICUBIC, POLICY_REFERENCE, POLICY_EXACT, POLICY_DOUBLE_PRECISION,
                   Request, Layout)
from .api import (AreaAverageInterpolation, AaiError, PinnedArray, make_request, query, resample_host, resample_batch_host, precision_check, resample_interleaved_host, resample_interleaved_device, resample_device, resample_multi_device,
                  band_source_rows, resample_band_device, synth_device, synth_rows_device, prepare, plan_shape, shutdown, debug_cell_min_waves, debug_skip_fixup, device_count, set_device, synchronize, last_kernel, last_error)

__all__ = ["AreaAverageInterpolation", "AaiError", "PinnedArray", "make_request", "query", "resample_host", "resample_batch_host", "precision_check", "resample_interleaved_host", "resample_interleaved_device", "resample_device", "resample_multi_device",
           "band_source_rows", "resample_band_device", "synth_device", "synth_rows_device", "prepare", "plan_shape", "shutdown", "debug_cell_min_waves", "debug_skip_fixup", "device_count", "set_device", "synchronize", "last_kernel", "last_error",
           "MODE_AREA", "MODE_FAST", "MODE_BILINEAR", "MODE_BICUBIC", "POLICY_REFERENCE", "POLICY_EXACT", "POLICY_DOUBLE_PRECISION",
           "Request", "Layout", "DTYPE_F32", "DTYPE_U8", "DTYPE_U16"]
