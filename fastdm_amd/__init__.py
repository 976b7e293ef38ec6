"""fastdm_amd: MI355X-native (gfx950) denoise-step hot path behind FastDM's operator / model_entry API.

Importing the package never touches the GPU; the HIP library is loaded lazily on the first kernel
call and a missing library is a hard error (there is no CPU or eager fallback in this package).
"""
__version__ = "0.1.0"
