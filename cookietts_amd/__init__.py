"""cookietts_amd - MI355X-native mel-to-wave hot path of CookiePPP/cookietts.

Python host mirrors of the reference's model API (``WaveGlow(**cfg).infer``) over a C-ABI
HIP library (``include/cookietts_hip.h``, sources in ``cookietts_amd/csrc``).
"""
from . import synthetic  # noqa: F401
from ._lib import set_f32_gemm_mode  # noqa: F401
from .alignment import alignment_metric, get_first_over_thresh  # noqa: F401
from .audio import STFT, Denoiser, TacotronSTFT  # noqa: F401
from .tacotron2 import Tacotron2, load_model  # noqa: F401
from .vocoder import WaveGlowVocoder, load_waveglow  # noqa: F401
from .waveglow import WaveGlow  # noqa: F401
from .waveglow_ax import WaveGlow as WaveFlow  # noqa: F401  (efficient_model_ax.WaveGlow, waveflow=True)

__all__ = ["set_f32_gemm_mode", "WaveGlow", "WaveGlowVocoder", "load_waveglow", "WaveFlow", "Tacotron2", "load_model", "STFT", "TacotronSTFT", "Denoiser", "alignment_metric",
           "get_first_over_thresh", "synthetic"]
